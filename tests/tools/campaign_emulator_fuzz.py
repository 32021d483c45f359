"""Ad-hoc (round 5): tests/fuzz_gpu.py's compress and decode cases on the emulator for a time budget, seed after seed, with the
hand-over rules forced (LBZAMD_HANDOVER0=1, LBZAMD_HANDOVER1=1: the rank rounds meet the closed runs) among them.  usage: campaign_emulator_fuzz.py first_seed seconds"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("LBZ_EMU_CHECK_SITES", "2")
os.environ.setdefault("LBZAMD_DWIDE", "0")
import test_emu_kernels as T
import fuzz_gpu
emu = T.Library(os.path.join(T.EMU_DIR, "_build", "liblbzamd_emu_1024.so"))
L = T.L
t0 = time.time()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
total = 0
while time.time() - t0 < budget:
    for knob in ({}, {"LBZAMD_HANDOVER0": "1"}, {"LBZAMD_HANDOVER1": "1"}):
        for k, v in knob.items(): os.environ[k] = v
        bad = fuzz_gpu.run(emu, L.orc_compress, seed, 60, small=True, save="/tmp")
        badd = fuzz_gpu.run_decode(emu, seed, 25, small=True)
        for k in knob: del os.environ[k]
        total += 85
        if bad or badd:
            print("FAIL seed", seed, knob, bad, badd, flush=True)
    print("seed", seed, "ok; cases so far", total, "t=%.0f" % (time.time() - t0), flush=True)
    seed += 1
print("done", total)
