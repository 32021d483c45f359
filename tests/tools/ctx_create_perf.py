"""Ad-hoc: what a context costs to create -- alone, several at once from threads, one after another."""
import sys, time, threading, ctypes as C
sys.path.insert(0, "/root/repo")
import lbzip2_amd
lib = lbzip2_amd.library(); L = lib.lib
L.lbzamd_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint, C.c_uint, C.c_uint]
L.lbzamd_destroy.argtypes = [C.c_void_p]
def create(slabs, out, i):
    h = C.c_void_p(); t = time.time(); rc = L.lbzamd_create(C.byref(h), -1, 9, slabs, 0); out[i] = (rc, time.time() - t, h)
def run(name, slabs, n, serial=False):
    out = [None] * n; t = time.time()
    if serial:
        for i in range(n): create(slabs, out, i)
    else:
        th = [threading.Thread(target=create, args=(slabs, out, i)) for i in range(n)]
        [x.start() for x in th]; [x.join() for x in th]
    wall = time.time() - t
    print(f"{name:40s} slabs {slabs:5d} x{n} {'serial' if serial else 'threads'}: wall {wall:.3f} s, each {[round(o[1], 3) for o in out]} rc {[o[0] for o in out]}", flush=True)
    t = time.time()
    for o in out:
        if o[2]: L.lbzamd_destroy(o[2])
    print(f"    destroy {time.time() - t:.3f} s", flush=True)
create(16, [None], 0)
for rep in range(2):
    run("one", 256, 1); run("one", 556, 1); run("one", 1112, 1)
    run("two at once", 256, 2); run("two one after the other", 256, 2, True)
    run("three at once", 371, 3); run("three one after the other", 371, 3, True)
    run("two at once", 556, 2); run("two one after the other", 556, 2, True)
