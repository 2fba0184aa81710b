"""Where the matrix pipe idles inside k_contract16d: the kernel's own phase trace (profiles/r06_contract_d_trace.txt).

Same method as tools/contract_trace.py (no thread-trace decoder, no PC sampling on this image): a profiling build
(`bash tools/build_variant.sh dtrace -DCONTRACT_D_TRACE kernels_posterior.hip bogp_api_sweep.hip`, copied over the package's library on the box)
stamps the shader clock in every wave of every workgroup: entry, loop start (first two k-pairs' operands requested), the start of EVERY k-pair,
loop end, after the drain, after the wave's own reduction, exit.  Each MFMA holds the pipe 64 cycles and the number of MFMAs of every k-pair is
known (8 x live column tiles), so every SIMD's time line (its two resident waves, one from each of the CU's two workgroups) can be replayed and
each idle cycle of the pipe booked on the phases the resident waves are in.

usage: python tools/contract_d_trace.py [C3] > profile.txt     (on the GPU box, with the trace build in place)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from bogp import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
w = bench.WORKLOADS[wl]
N, d, M = w["N"], w["d"], w["M"]
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
lib = _lib.load()
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(w["kernel"], _lib.MODE_NOISY, np.r_[np.full(d, w["theta"]), 0.9], 1e-6, False, 0.0)
Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
for _ in range(3):
    res = eng.sweep(w["acq"], float(y.min()), True)
print("# workload %s: N=%d d=%d M=%d; sweep result %s" % (wl, N, d, M, res))
print("# last_timing of the traced build:", eng.last_timing())

fn = lib.bogp_debug_contract_trace
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
used = C.c_size_t(0)
dims = (C.c_int * 4)()
assert fn(None, 0, C.byref(used), dims) == 0
nMt, nJ, NJ16, NST = [int(v) for v in dims]
buf = np.empty(used.value, dtype=np.uint64)
assert fn(buf.ctypes.data, used.value, C.byref(used), dims) == 0
tr = buf.reshape(nMt * nJ, 4, NST).astype(np.int64)
print("# trace: nMt=%d nJ=%d NJ16=%d words/wave=%d" % (nMt, nJ, NJ16, NST))

NR, NWJ = 4, 4
JT16 = NWJ * NR
hw = tr[:, :, 0]
hwid = hw & 0xffffffff
xcc = (hw >> 32) & 0xf
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
cu_key = simd_key >> 2
print("# distinct SIMDs seen: %d, distinct CUs: %d" % (len(np.unique(simd_key)), len(np.unique(cu_key))))
nkp_w = tr[:, :, 7] & 0xffffffff
nkp_full_w = tr[:, :, 7] >> 32
# s_memtime is per die: every CU's stamps are moved to that CU's own first entry stamp (no comparison below crosses a CU)
for k in np.unique(cu_key):
    m = cu_key == k
    base = tr[:, :, 1][m].min()
    sub = tr[m]
    sub[:, 1:7] -= base
    sub[:, 8:] -= base
    sub[:, 8:][sub[:, 8:] < 0] = 0
    tr[m] = sub
t_entry, t_exit = tr[:, :, 1], tr[:, :, 6]
T0, T1 = int(t_entry.min()), int(t_exit.max())
TT = T1 - T0
print("# kernel span by the stamps: %d cycles (shader clock); workgroups %d" % (TT, nMt * nJ))

TAGS = ["NONE", "SETUP", "FIRST", "FULL", "Z4", "Z3", "Z2", "Z1", "Z0", "DRAIN", "RED", "FINAL"]
TID = {t: i for i, t in enumerate(TAGS)}
ZT = {4: TID["Z4"], 3: TID["Z3"], 2: TID["Z2"], 1: TID["Z1"], 0: TID["Z0"]}


def wave_tiles(jg, wv):
    return [jg * JT16 + ((NWJ * (ni + 1) - 1 - wv) if (ni & 1) else (NWJ * ni + wv)) for ni in range(NR)]


waves = []
rows = {t: [0, 0, 0.0] for t in TAGS}
full_len = []
zone_len = {n: [] for n in range(5)}
for jg in range(nJ):
    g0 = (nJ - 1 - jg) * nMt
    nkp_full = 2 * jg * JT16
    for wv in range(4):
        nkp = int(nkp_w[g0, wv])
        jt = [j if j < NJ16 else -1 for j in wave_tiles(jg, wv)]
        # intervals: entry -> loop start (SETUP), k-pair 0 (FIRST: includes the wait for the first operands), k-pairs 1 .. nkp-1, loop end -> drain (DRAIN),
        # -> wave reduction done (RED), -> exit (FINAL: workgroup barrier + the four-wave sum + store)
        idx = [1, 2] + [8 + kp for kp in range(1, nkp)] + [3, 4, 5, 6]
        tg, mf = [TID["SETUP"]], [0.0]
        for kp in range(nkp):
            act = sum(1 for j in jt if j >= 0 and (kp < nkp_full or (kp >> 1) <= j))
            tag = TID["FIRST"] if kp == 0 else (TID["FULL"] if kp < nkp_full else ZT[act])
            tg.append(tag); mf.append(8.0 * act * 64)
        tg += [TID["DRAIN"], TID["RED"], TID["FINAL"]]; mf += [0.0, 0.0, 0.0]
        # (the stamp of k-pair kp sits in front of its loads; k-pair 0's interval starts at the loop-start stamp)
        idx = np.asarray(idx); tg = np.asarray(tg, dtype=np.int8); mf = np.asarray(mf)
        ST = tr[g0:g0 + nMt, wv][:, idx]
        LN = np.maximum(np.diff(ST, axis=1), 1)
        for t in range(len(TAGS)):
            m = tg == t
            if m.any():
                rows[TAGS[t]][0] += int(m.sum()) * nMt; rows[TAGS[t]][1] += int(LN[:, m].sum()); rows[TAGS[t]][2] += float(mf[m].sum()) * nMt
        mfull = tg == TID["FULL"]
        if mfull.any():
            full_len.append(LN[:, mfull].ravel())
        for n in range(5):
            mz = tg == ZT[n]
            if mz.any():
                zone_len[n].append(LN[:, mz].ravel())
        for i in range(nMt):
            waves.append((int(simd_key[g0 + i, wv]), ST[i], tg, mf / LN[i]))

nsimd = len(np.unique(simd_key))
print("\n== per phase, summed over all waves: intervals, mean length (cycles), MFMA pipe cycles the wave itself needs in it, own / length, share of all wave time")
print("   (a FULL k-pair = 32 MFMAs = 2048 pipe cycles; with two waves a SIMD the fair share of wall time is 4096)")
for t in TAGS[1:]:
    n, L, m = rows[t]
    if n:
        print("   %-5s n=%9d  mean len %8.1f  own MFMA cycles %8.1f  own/len %.3f   share %.4f" % (t, n, L / n, m / n, m / max(L, 1), L / (TT * 2.0 * nsimd)))
fl = np.concatenate(full_len)
print("   FULL k-pair length: p05 %.0f p25 %.0f median %.0f p75 %.0f p95 %.0f p99 %.0f max %.0f" % tuple(np.percentile(fl, [5, 25, 50, 75, 95, 99, 100])))
for n in (4, 3, 2, 1, 0):
    if zone_len[n]:
        z = np.concatenate(zone_len[n])
        print("   zone k-pair with %d live tiles (%4d pipe cycles): n=%8d mean %7.1f median %7.0f p95 %7.0f" % (n, 512 * n, len(z), z.mean(), np.median(z), np.percentile(z, 95)))

STEP = 251
ts = np.arange(T0, T1, STEP, dtype=np.int64)
ns = len(ts)
keys = np.unique(simd_key)
kidx = {int(k): i for i, k in enumerate(keys)}
tagA = np.zeros((len(keys), ns), dtype=np.int8); tagB = np.zeros((len(keys), ns), dtype=np.int8)
rho = np.zeros((len(keys), ns), dtype=np.float32)
nres = np.zeros((len(keys), ns), dtype=np.int8)
t0 = time.time()
for (k, st, tg, rh) in waves:
    i = kidx[k]
    a, b = np.searchsorted(ts, [st[0], st[-1]])
    if b <= a:
        continue
    seg = np.clip(np.searchsorted(st, ts[a:b], side="right") - 1, 0, len(tg) - 1)
    first = nres[i, a:b] == 0
    tA = tagA[i, a:b]; tB = tagB[i, a:b]
    tA[first] = tg[seg][first]
    tB[~first] = tg[seg][~first]
    rho[i, a:b] += rh[seg]
    nres[i, a:b] += 1
print("\n# sampled %d SIMDs x %d samples in %.1f s" % (len(keys), ns, time.time() - t0))
idle = np.clip(1.0 - rho, 0.0, 1.0)
over = np.clip(rho - 1.0, 0.0, None)
print("== MFMA pipe by the trace: busy %.4f, idle %.4f (demand above 1.0 in a sample, i.e. interval granularity error: %.4f)" % (1 - idle.mean(), idle.mean(), over.mean()))
print("   resident waves per SIMD sample: 0: %.4f  1: %.4f  2: %.4f  >2: %.4f" % (tuple((nres == v).mean() for v in (0, 1, 2)) + ((nres > 2).mean(),)))
pair = np.zeros((len(TAGS), len(TAGS)))
lo = np.minimum(tagA, tagB); hi = np.maximum(tagA, tagB)
np.add.at(pair, (lo.ravel(), hi.ravel()), idle.ravel())
one = np.zeros(len(TAGS))
for a in range(len(TAGS)):
    for b in range(a, len(TAGS)):
        one[a] += pair[a, b] / 2; one[b] += pair[a, b] / 2
print("\n== idle MFMA-pipe time by the phase of the resident waves (half to each of the two; NONE = empty wave slot), as % of ALL SIMD time")
for t in np.argsort(-one):
    if one[t] > 0:
        print("   %-6s %6.3f %%" % (TAGS[t], 100.0 * one[t] / idle.size))
print("   total  %6.3f %%" % (100.0 * idle.sum() / idle.size))
print("\n== the same by PAIR of phases (top 20)")
flat = [(pair[a, b], TAGS[a], TAGS[b]) for a in range(len(TAGS)) for b in range(a, len(TAGS)) if pair[a, b] > 0]
for v, a, b in sorted(flat, reverse=True)[:20]:
    print("   %-6s + %-6s %6.3f %%" % (a, b, 100.0 * v / idle.size))
print("\n== idle share along the launch (tenths of the span)")
for q in range(10):
    s0, s1 = q * ns // 10, (q + 1) * ns // 10
    print("   %d0-%d0 %%: idle %.4f, empty slots %.4f" % (q, q + 1, idle[:, s0:s1].mean(), 1 - nres[:, s0:s1].mean() / 2))
gaps = []
by = {}
for (k, st, tg, rh) in waves:
    by.setdefault(k, []).append((int(st[0]), int(st[-1])))
for k, lst in by.items():
    lst.sort()
    ends = []
    for s, e in lst:
        cand = [x for x in ends if x <= s]
        if cand:
            x = max(cand); gaps.append(s - x); ends.remove(x)
        ends.append(e)
gaps = np.asarray(gaps)
print("\n== workgroup turnover: gap between a wave's exit stamp and the next wave's entry stamp in the same SIMD slot: n=%d mean %.0f median %.0f p90 %.0f cycles" % (len(gaps), gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
pro = np.asarray([st[2] - st[0] for (_, st, _, _) in waves])
print("   entry -> end of the first k-pair (set-up + the first operands' round trip to HBM + 32 MFMAs): mean %.0f median %.0f p90 %.0f cycles" % (pro.mean(), np.median(pro), np.percentile(pro, 90)))
epi = np.asarray([st[-1] - st[-4] for (_, st, _, _) in waves])
print("   last k-pair's end -> exit (drain + wave reduction + workgroup barrier + store): mean %.0f median %.0f cycles" % (epi.mean(), np.median(epi)))
# skew of the four waves of a workgroup (they never wait for each other before the epilogue): spread of their loop-end stamps
le = tr[:, :, 3]
print("   spread of the four waves' loop-end stamps inside a workgroup: mean %.0f median %.0f p90 %.0f cycles (the last one holds the workgroup's barrier)" % (
    (le.max(axis=1) - le.min(axis=1)).mean(), np.median(le.max(axis=1) - le.min(axis=1)), np.percentile(le.max(axis=1) - le.min(axis=1), 90)))
