bash profiles/microbench/r05_final_pmc.sh c5 markov laplace2d banded shell laplace3d
