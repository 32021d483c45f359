"""Ad-hoc: what the threads of a running process are doing -- samples /proc/PID/task/*/{stat,syscall,comm} for a while and
prints, per thread name, the CPU time used and a histogram of the system calls the samples found the threads in.
usage: sample_threads.py PID [seconds]"""
import os, sys, time, collections
pid = int(sys.argv[1]); dur = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
SYS = {202: "futex", 24: "sched_yield", 9: "mmap", 11: "munmap", 16: "ioctl", 7: "poll", 230: "clock_nanosleep", 35: "nanosleep", 28: "madvise", 10: "mprotect", 0: "read", 1: "write", 232: "epoll_wait", 271: "ppoll", 12: "brk"}
errs = collections.Counter(); hist = collections.Counter(); states = collections.Counter(); t_end = time.time() + dur
first = {}; last = {}
while time.time() < t_end:
    try:
        tids = os.listdir(f"/proc/{pid}/task")
    except FileNotFoundError:
        break
    for t in tids:
        try:
            st = open(f"/proc/{pid}/task/{t}/stat").read().rsplit(")", 1)[1].split()
            comm = open(f"/proc/{pid}/task/{t}/comm").read().strip()
        except Exception as e:
            errs[repr(e)[:60]] += 1
            continue
        try:
            sc = open(f"/proc/{pid}/task/{t}/syscall").read().split()
        except Exception as e:
            errs[repr(e)[:60]] += 1
            try:
                sc = ["wchan:" + open(f"/proc/{pid}/task/{t}/wchan").read().strip()]
            except Exception:
                sc = []
        state, ut, stt = st[0], int(st[11]), int(st[12])
        first.setdefault(t, (ut, stt)); last[t] = (ut, stt, comm)
        states[state] += 1
        if sc and sc[0].startswith("wchan:"):
            hist[(state, sc[0])] += 1
        elif sc and sc[0] not in ("running", "-1"):
            hist[(state, SYS.get(int(sc[0]), sc[0]))] += 1
        elif sc:
            hist[(state, sc[0])] += 1
    time.sleep(0.003)
print("errors", dict(errs)); print("states", dict(states))
print("syscalls (state, call): samples", sorted(hist.items(), key=lambda kv: -kv[1])[:14])
by = collections.Counter(); bys = collections.Counter()
for t, (ut, stt, comm) in last.items():
    by[comm] += ut - first[t][0]; bys[comm] += stt - first[t][1]
print("ticks by thread name (user, sys):", {k: (by[k], bys[k]) for k in by if by[k] + bys[k] > 0})
