import sys, cProfile, pstats, time, os
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--emulate-ranks", "8", "--emulate-rank", "4", "--steps", "300", "--warmup", "20", "--no-cpu-baseline", "--no-bandwidth", "--no-pmc", "--profile-steps", "0"]
import bench
pr = cProfile.Profile(); pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)
st.print_callers("contiguous")
st.print_callers("camera_on_device")
