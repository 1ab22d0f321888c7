import sys, os
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_conv.py")).read().split("for split in (0, 1):")[0])
run("scale0 C256 L4500 k5", 64, 256, 4500, 5, 2, 1, 1, 1, reps=5)
