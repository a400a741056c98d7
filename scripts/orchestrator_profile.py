import cProfile, pstats, os, sys, io
sys.argv = ["orchestrator_bench.py"]
os.environ["DEVICE_ONLY"] = "1"
os.environ.setdefault("SIDE", "128")
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orchestrator_bench.py")).read()
# profile only the second (timed) run: wrap oi.run
src = src.replace("        tabs = oi.run(store_path=os.path.join(d, \"s\"), store_every=se, engine_chunk=chunk)",
 "        pr = cProfile.Profile(); pr.enable()\n        tabs = oi.run(store_path=os.path.join(d, \"s\"), store_every=se, engine_chunk=chunk)\n        pr.disable(); s_ = io.StringIO(); pstats.Stats(pr, stream=s_).sort_stats('cumulative').print_stats(45); print(s_.getvalue())")
exec(compile(src, "orchestrator_bench.py", "exec"))
