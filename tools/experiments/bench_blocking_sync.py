"""EXPERIMENT: bench.py with the host's wait for a stream BLOCKING (hipDeviceScheduleBlockingSync, set before the device is first
used) instead of the runtime's default (spinning when the host has more cores than devices) - does a host with few free cores
(taskset -c 0-3) keep the 8-lane rate?   python tools/experiments/bench_blocking_sync.py [bench.py arguments]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipSetDeviceFlags(ctypes.c_uint(int(os.environ.get("DS_EXP_SCHED_FLAG", "4"))))  # 4: hipDeviceScheduleBlockingSync, 2: Yield, 1: Spin
print("hipSetDeviceFlags(blocking sync) ->", rc, file=sys.stderr)
sys.argv = ["bench.py"] + sys.argv[1:]
import bench  # noqa: E402
bench.main()
