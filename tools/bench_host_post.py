"""Host passes behind the kernels at cohort scale: merging the window calls of one variant (unifyCalls) and formatting
its VCF record. python tools/bench_host_post.py [samples ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
lib.pdh_bench_merge_and_format.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
for S in [int(a) for a in sys.argv[1:]] or [1000, 10000, 50000]:
    out = (C.c_double * 3)()
    windows, records = 34, 20
    assert lib.pdh_bench_merge_and_format(S, windows, records, out) == 0
    print("samples %6d: merge of %d window calls %.4f s (%.0f ns per window x sample); one record of %.1f MB formatted in %.4f s "
          "(%.0f ns per sample column, %.0f MB/s)" % (S, windows, out[0], out[0] / (windows * S) * 1e9, out[2] / 1e6, out[1] / records,
                                                       out[1] / records / S * 1e9, out[2] / 1e6 / (out[1] / records)))
