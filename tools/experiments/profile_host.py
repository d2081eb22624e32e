"""Host-side profile of bench.py's step loop (where a host-bound step - configs[4], 0.55 ms - spends its Python time).
usage (GPU box): python tools/experiments/profile_host.py <bench args>   -> top functions by own time on stdout"""
import cProfile
import pstats
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.argv = ["bench.py"] + sys.argv[1:]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(45)
