"""Counterpart of the reference's experiments/tabulators/tabulate_toy.py: error statistics (KL, Bures, mean / variance
error, skewness, kurtosis) of the result files examples/toy_{gibbs,filter,pmcmc,twisted}.py write.

    python examples/tabulate_toy.py results/gibbs-eb-const-100-*.npz
"""
import glob
import sys

from fbs_amd.metrics import tabulate

if __name__ == "__main__":
    files = sorted(f for pat in sys.argv[1:] for f in glob.glob(pat))
    if not files:
        raise SystemExit("usage: tabulate_toy.py <result .npz files>")
    stats = tabulate(files)
    print(f"{len(files)} run(s) | " + " | ".join(f"{k} {m:.4f} {s:.4f}" for k, (m, s) in stats.items()))
