import ctypes as C, os, sys, subprocess, runpy
# run a bench tool in-process with the profiling library, then read the phase counters
sys.argv = sys.argv[1:]
import atexit
def dump():
    lib = C.CDLL(os.environ["GDIET_HIP_LIB"])
    a = (C.c_ulonglong * 8)()
    lib.gdiet_hip_debug_seed_prof(a)
    tot = sum(a[:6]) or 1
    print("seed phases (wall_clock64 ticks, summed over wavefronts): S1 sketch2 %.1f%%  S3 shift %.1f%%  S2 sketch3 %.1f%%  S4 flt %.1f%%  S5 probes %.1f%%  S5 select+copy %.1f%%   total %d | inside the sketches: slices %.1f%%, scan+copy %.1f%%" % tuple([100.0 * a[i] / sum(a[:6]) for i in range(6)] + [sum(a[:6]), 100.0 * a[6] / sum(a[:6]), 100.0 * a[7] / sum(a[:6])]), file=sys.stderr)
atexit.register(dump)
runpy.run_path(sys.argv[0], run_name="__main__")
