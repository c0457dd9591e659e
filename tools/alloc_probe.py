import sys, time, ctypes
sys.path.insert(0, ".")
from halo2_vectordb_amd import api
from halo2_vectordb_amd._lib import check
api.init(0)
lib = api.init()
def t(f):
    t0 = time.perf_counter(); r = f(); return r, round(time.perf_counter() - t0, 3)
for gb, touch in ((90, False), (90, True), (40, True), (10, True)):
    b, ta = t(lambda: api.DeviceBuffer(gb << 30))
    tm = 0
    if touch:
        _, tm = t(lambda: (check(lib.vdb_memset_dev(b.ptr, 0, ctypes.c_size_t(gb << 30))), api.sync()))
    _, tf = t(lambda: b.free())
    b2, ta2 = t(lambda: api.DeviceBuffer(1 << 30))
    _, tf2 = t(lambda: b2.free())
    b3, ta3 = t(lambda: api.DeviceBuffer(gb << 30))
    _, tf3 = t(lambda: b3.free())
    print(f"{gb} GB touch={touch}: malloc {ta} s, memset {tm} s, free {tf} s, next malloc(1 GB) {ta2} s, its free {tf2} s, malloc same size again {ta3} s, free {tf3}", flush=True)
