"""Development aid: where the host (Python) time of a bench step goes (run on the GPU box)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.argv = ["bench.py", "--steps", "60", "--warmup", "5", "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
