"""Decode the work counters of a PHMRF_PHASE_CLOCK build of strip_cols_kernel (tools/trace.py output on stdin)."""
import sys, ast
l = [x for x in sys.stdin.read().strip().split("\n") if x.startswith("work:")][-1]
d = ast.literal_eval(l[l.index("{"):])
pairs = float(sys.argv[1]) if len(sys.argv) > 1 else 3298310.0
print("general sweeps per pair %.2f" % ((d["units"] - pairs) / 4096 / pairs))
ph = {"extraction": d["cells"] - 51.9e6, "ids+staging": d["staged_cells"] - 75e6, "single-site costs": d["dp_steps"] - 32e6,
      "DP": d["swept_cells"], "sweeps": d["label_cells"]}
t = sum(ph.values())
for k, v in ph.items():
    print("%-18s %5.0f M  %5.1f %%   cycles/pair %6.0f" % (k, v / 1e6, 100 * v / t, v * 16 / pairs))
