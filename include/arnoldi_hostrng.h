/* Host-only helper of the drop-in (no device code, no HIP types): NumPy's legacy standard-normal stream.
 *
 * Replaces, for large n, the single call the reference's start vector consists of --
 * np.random.randn(n) in rand_normalized_vector (/root/reference/src/arnoldi/utils.py:7-13, called from
 * krylov_schur.py:45-46) -- with the same bits computed faster (the Mersenne Twister sequentially, the polar-method
 * arithmetic on host threads).  Exported by libarnoldi_hip.so; bound in arnoldi_amd/utils.py. */
#ifndef ARNOLDI_HOSTRNG_H
#define ARNOLDI_HOSTRNG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* out[0..n) = what np.random.randn(n) returns for a legacy RandomState whose state is (key[624], *pos, *has_gauss, *gauss)
 * -- the tuple np.random.get_state() gives; on return the four describe the state NumPy would be in after that call
 * (write them back with np.random.set_state).  Returns 0 on success, non-zero on a bad argument or an internal error
 * (the state is then unspecified: the caller restores its saved copy and falls back to NumPy). */
int aks_legacy_randn(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, double *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
