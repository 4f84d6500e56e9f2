"""Version of the MI355X drop-in (independent of the reference's own version number)."""
ABI = 6                      # must equal AKS_ABI_VERSION of include/arnoldi_hip.h (checked against the loaded library by _hip.load)
__version__ = f"0.3.0+gfx950.abi{ABI}"
