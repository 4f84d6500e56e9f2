bash profiles/microbench/r05_final_pmc.sh banded shell laplace3d
