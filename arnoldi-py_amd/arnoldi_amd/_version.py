"""Version of the MI355X drop-in (independent of the reference's own version number)."""
ABI = 1                      # must equal AKS_ABI_VERSION of include/arnoldi_hip.h
__version__ = f"0.1.0+gfx950.abi{ABI}"
