"""Where the matrix pipe idles inside k_contract16<4>: an instruction-phase trace the kernel keeps itself (profiles/r06_contract_att.txt).

rocprofv3 --att needs a trace decoder this image does not ship, and the driver of the GPU boxes offers no PC-sampling configuration
(gpurun_out/r06_pcsamp*/: both attempts recorded in the profile's header) -- so a profiling build (`bash tools/build_variant.sh trace
-DCONTRACT_TRACE kernels_posterior.hip bogp_api.hip`, copied over the package's library on the box) stamps the shader clock (s_memtime) in every
wave of every workgroup: kernel entry, end of the prologue, and per 32-row block the barrier exit, the top of k-pairs 1..3, the last MFMA's
issue, the stage store; then the drain, the two reduction phases and the exit.  Each wave also records HW_ID / XCC_ID, so the two waves
that share a SIMD (one from each of the CU's two workgroups) are put on ONE time axis.  The MFMA count of every interval is known
(8 x live column tiles a k-pair), each MFMA holds the pipe 64 cycles; the SIMD is sampled every 251 cycles and the idle share of a
sample, 1 - sum(MFMA cycles / interval length) over the resident waves, is booked on the phases the resident waves are in.

usage: python tools/contract_trace.py [C3] > profile.txt     (on the GPU box, with the trace build in place)"""
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
tm = eng.last_timing()
print("# workload %s: N=%d d=%d M=%d; sweep result %s" % (wl, N, d, M, res))
print("# last_timing of the traced build:", tm)

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
simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)  # [wg][wave]
cu_key = simd_key >> 2
print("# distinct SIMDs seen: %d, distinct CUs: %d, XCCs: %s, SEs: %s" % (len(np.unique(simd_key)), len(np.unique(cu_key)), np.unique(xcc), np.unique(se)))
# s_memtime is not one counter for the device (the raw values of different dies are ~1e12 apart): every CU's stamps are moved to that CU's own
# first entry stamp -- the CUs start a launch within microseconds of each other, and no comparison below crosses a CU
raw_entry = tr[:, :, 1].copy()
for x in np.unique(xcc):
    mx = xcc == x
    print("#   XCC %d raw entry stamps: min %d, spread over its SEs (min per SE - die min): %s" % (
        x, raw_entry[mx].min(), [int(raw_entry[mx & (se == e)].min() - raw_entry[mx].min()) for e in np.unique(se[mx])]))
for k in np.unique(cu_key):
    m = cu_key == k
    base = tr[:, :, 1][m].min()
    sub = tr[m]
    sub[:, 1:] -= base
    sub[:, 1:][sub[:, 1:] < 0] = 0   # never-written slots (blocks beyond the group's last)
    tr[m] = sub
t_entry = tr[:, :, 1]
t_exit = tr[:, :, 6]
spans = np.asarray([t_exit[cu_key == k].max() for k in np.unique(cu_key)])
print("# per-CU span (first entry -> last exit): min %d median %d max %d ticks" % (spans.min(), np.median(spans), spans.max()))
T0 = int(t_entry.min()); T1 = int(t_exit.max())
TT = T1 - T0
print("# kernel span by the stamps: %d cycles (shader clock); workgroups %d" % (TT, nMt * nJ))

# ---- per-wave interval lists ------------------------------------------------------------------------------------------------------
TAGS = ["NONE", "PRO", "BAR", "FULL", "Z4", "Z3", "Z2", "Z1", "Z0", "STS", "DRAIN", "EPI1", "EPI2"]
TID = {t: i for i, t in enumerate(TAGS)}
ZT = {4: TID["Z4"], 3: TID["Z3"], 2: TID["Z2"], 1: TID["Z1"], 0: TID["Z0"]}


def wave_tiles(jg, wv):
    return [jg * JT16 + ((NWJ * (ni + 1) - 1 - wv) if (ni & 1) else (NWJ * ni + wv)) for ni in range(NR)]


waves = []  # (simd_key, stamps[], tags[], rho_cycles[])
blk_rows = []  # block-level records: (jg, zone index or -1, block length, mfma cycles of the wave, stage-store wait, barrier wait, issue span)
kp_rows = {t: [0, 0, 0] for t in TAGS}  # per tag: intervals, total length, total mfma cycles
for jg in range(nJ):
    g0 = (nJ - 1 - jg) * nMt
    kmax16 = min((jg + 1) * JT16, NJ16)
    nkb = min(kmax16 >> 1, 64)
    nkb_full = jg * (JT16 // 2)
    for wv in range(4):
        jt = [j if j < NJ16 else -1 for j in wave_tiles(jg, wv)]
        idx = [1, 2]; tg = [TID["PRO"], TID["PRO"]]; mf = [0, 0]
        btot = []
        for kb in range(nkb):
            guarded = kb >= nkb_full
            tot = 0
            for s in range(4):
                kb16 = (kb * 4 + s) >> 1
                act = sum(1 for j in jt if j >= 0 and (not guarded or kb16 <= j))
                tot += 8 * act
                idx.append(8 + 8 * kb + s); tg.append(TID["FULL"] if not guarded else ZT[act]); mf.append(8 * act * 64)
            idx.append(8 + 8 * kb + 4); tg.append(TID["STS"]); mf.append(0)
            idx.append(8 + 8 * kb + 5); tg.append(TID["BAR"]); mf.append(0)
            btot.append(tot * 64)
        idx += [3, 4, 5, 6]; tg += [TID["DRAIN"], TID["EPI1"], TID["EPI2"]]; mf += [0, 0, 0]
        idx = np.asarray(idx); tg = np.asarray(tg, dtype=np.int8); mf = np.asarray(mf, dtype=np.float64)
        ST = tr[g0:g0 + nMt, wv][:, idx]                      # (nMt, L + 1)
        LN = np.maximum(np.diff(ST, axis=1), 1)               # (nMt, L)
        for t in range(len(TAGS)):
            m = tg == t
            if m.any():
                kp_rows[TAGS[t]][0] += int(m.sum()) * nMt; kp_rows[TAGS[t]][1] += int(LN[:, m].sum()); kp_rows[TAGS[t]][2] += float(mf[m].sum()) * nMt
        for kb in range(nkb):
            o = 2 + 6 * kb
            s0 = ST[:, o]; e4 = ST[:, o + 4]; s5 = ST[:, o + 5]; nxt = ST[:, o + 6]
            z = kb - nkb_full if kb >= nkb_full else -1
            blk_rows.append(np.stack([np.full(nMt, jg), np.full(nMt, z), nxt - s0, np.full(nMt, btot[kb]), s5 - e4, nxt - s5, e4 - s0], axis=1))
        for i in range(nMt):
            waves.append((int(simd_key[g0 + i, wv]), ST[i], tg, mf / LN[i], g0 + i, wv))
blk_rows = np.concatenate(blk_rows, axis=0)

print("\n== per phase, summed over all waves: intervals, mean length (cycles), MFMA pipe cycles the wave itself needs in it, ratio")
print("   (a k-pair of a FULL block = 32 MFMAs = 2048 pipe cycles; with two waves a SIMD the fair share of wall time is 4096)")
for t in TAGS[1:]:
    n, L, m = kp_rows[t]
    if n:
        print("   %-5s n=%9d  mean len %8.1f  own MFMA cycles %8.1f  own/len %.3f   total len share %.4f" % (t, n, L / n, m / n, m / max(L, 1), L / (TT * 4.0 * len(np.unique(simd_key)) / 4 * 2)))

br = blk_rows.astype(np.float64)
print("\n== blocks (per wave): mean wall cycles of a 32-row block, the wave's own MFMA cycles in it, MFMA-issue span, stage-store wait, barrier wait")
m = br[:, 1] < 0
print("   FULL blocks      n=%8d  len %8.1f  own MFMA %7.1f  s0->last MFMA %8.1f  stage store %7.1f  barrier %7.1f" % (m.sum(), br[m, 2].mean(), br[m, 3].mean(), br[m, 6].mean(), br[m, 4].mean(), br[m, 5].mean()))
for z in range(8):
    m = br[:, 1] == z
    if m.any():
        print("   zone block %d     n=%8d  len %8.1f  own MFMA %7.1f  s0->last MFMA %8.1f  stage store %7.1f  barrier %7.1f" % (z, m.sum(), br[m, 2].mean(), br[m, 3].mean(), br[m, 6].mean(), br[m, 4].mean(), br[m, 5].mean()))
for jg in range(nJ):
    m = (br[:, 0] == jg) & (br[:, 1] >= 0)
    m2 = (br[:, 0] == jg) & (br[:, 1] < 0)
    print("   column group %d: zone blocks mean len %8.1f (own MFMA %7.1f); full blocks mean len %8.1f" % (jg, br[m, 2].mean(), br[m, 3].mean(), br[m2, 2].mean() if m2.any() else float("nan")))

# ---- SIMD sampling ----------------------------------------------------------------------------------------------------------------
STEP = 251
ts = np.arange(T0, T1, STEP, dtype=np.int64)
ns = len(ts)
keys = np.unique(simd_key)
kidx = {int(k): i for i, k in enumerate(keys)}
tagA = np.zeros((len(keys), ns), dtype=np.int8); tagB = np.zeros((len(keys), ns), dtype=np.int8)
rho = np.zeros((len(keys), ns), dtype=np.float32)
nres = np.zeros((len(keys), ns), dtype=np.int8)
t0 = time.time()
for (k, st, tg, rh, g, wv) in waves:
    i = kidx[k]
    a, b = np.searchsorted(ts, [st[0], st[-1]])
    if b <= a:
        continue
    seg = np.searchsorted(st, ts[a:b], side="right") - 1
    seg = np.clip(seg, 0, len(tg) - 1)
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
tot_idle = idle.sum()
pair = np.zeros((len(TAGS), len(TAGS)))
lo = np.minimum(tagA, tagB); hi = np.maximum(tagA, tagB)
np.add.at(pair, (lo.ravel(), hi.ravel()), idle.ravel())
one = np.zeros(len(TAGS))
for a in range(len(TAGS)):
    for b in range(a, len(TAGS)):
        one[a] += pair[a, b] / 2; one[b] += pair[a, b] / 2
print("\n== idle MFMA-pipe time by the phase of the resident waves (half to each of the two; NONE = empty wave slot), as %% of ALL SIMD time")
for t in np.argsort(-one):
    if one[t] > 0:
        print("   %-6s %6.3f %%" % (TAGS[t], 100.0 * one[t] / idle.size))
print("   total  %6.3f %%" % (100.0 * tot_idle / idle.size))
print("\n== the same by PAIR of phases (top 25)")
flat = [(pair[a, b], TAGS[a], TAGS[b]) for a in range(len(TAGS)) for b in range(a, len(TAGS)) if pair[a, b] > 0]
for v, a, b in sorted(flat, reverse=True)[:25]:
    print("   %-6s + %-6s %6.3f %%" % (a, b, 100.0 * v / idle.size))
# time axis: idle share per tenth of the launch
print("\n== idle share along the launch (tenths of the span)")
for q in range(10):
    s0, s1 = q * ns // 10, (q + 1) * ns // 10
    print("   %d0-%d0 %%: idle %.4f, empty slots %.4f" % (q, q + 1, idle[:, s0:s1].mean(), 1 - nres[:, s0:s1].mean() / 2))
# turnover: gap between a wave's exit and the next wave's entry on the same SIMD slot
gaps = []
by = {}
for (k, st, tg, rh, g, wv) in waves:
    by.setdefault(k, []).append((int(st[0]), int(st[-1])))
for k, lst in by.items():
    lst.sort()
    # two slots: greedy assignment
    ends = []
    for s, e in lst:
        cand = [x for x in ends if x <= s]
        if cand:
            x = max(cand); gaps.append(s - x); ends.remove(x)
        ends.append(e)
gaps = np.asarray(gaps)
print("\n== workgroup turnover: gap between a wave's exit stamp and the next wave's entry stamp in the same SIMD slot: n=%d mean %.0f median %.0f p90 %.0f cycles" % (len(gaps), gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
pro = np.asarray([st[2] - st[0] for (_, st, _, _, _, _) in waves])
print("   entry -> first block's barrier exit (prologue: first B fragments + tile 0 from HBM): mean %.0f median %.0f p90 %.0f cycles" % (pro.mean(), np.median(pro), np.percentile(pro, 90)))
epi = np.asarray([st[-1] - st[-4] for (_, st, _, _, _, _) in waves])
print("   last MFMA issued -> exit (drain + reduction + store): mean %.0f median %.0f cycles" % (epi.mean(), np.median(epi)))

# ---- barrier skew: who arrives last, and does the same wave stay last? --------------------------------------------------------------
print("\n== barrier skew inside a workgroup (column group 7, full blocks): arrival = stage-store stamp of the four waves")
g0 = 0  # column group nJ-1 is launched first: workgroups 0 .. nMt-1
nkb_full7 = (nJ - 1) * (JT16 // 2)
arr = np.stack([tr[g0:g0 + nMt, wv, 8 + 8 * np.arange(min(nkb_full7, 56)) + 5] for wv in range(4)], axis=2)   # (nMt, blocks, 4)
rel = arr - arr.min(axis=2, keepdims=True)
last = rel.argmax(axis=2)
print("   skew (last arrival - first arrival): mean %.0f median %.0f p90 %.0f ticks; mean wait of a wave (last - own): %.0f" % (
    rel.max(axis=2).mean(), np.median(rel.max(axis=2)), np.percentile(rel.max(axis=2), 90), (rel.max(axis=2, keepdims=True) - rel).mean()))
print("   wave that arrives last: %s" % ["w%d %.3f" % (i, (last == i).mean()) for i in range(4)])
print("   same wave last in consecutive blocks: %.3f (0.25 = independent)" % (last[:, 1:] == last[:, :-1]).mean())
sid = simd[g0:g0 + nMt]
print("   wave -> SIMD map of the first workgroups: %s" % [list(map(int, sid[i])) for i in range(4)])
# drift of a wave against its workgroup over the blocks: cumulative lead at the barrier, per wave, per workgroup -> is a wave persistently slow?
lead = (rel.max(axis=2, keepdims=True) - rel)  # wait per wave per block
per_wave_mean = lead.mean(axis=1)             # (nMt, 4)
print("   per-workgroup mean wait of its waves, sorted within the workgroup (least .. most waiting): %s" % np.sort(per_wave_mean, axis=1).mean(axis=0).round(0))
# the k-pair speed of a wave while its partner (the other workgroup's wave on the SIMD) is inside a barrier / stage phase is not directly stamped;
# proxy: a FULL k-pair's length distribution (2048 = alone on the pipe at full rate, 4096 = fair share with a partner in FULL)
kl = []
for wv in range(4):
    b = tr[g0:g0 + nMt, wv, 8:8 + 8 * min(nkb_full7, 56)].reshape(nMt, -1, 8)
    kl.append(np.diff(b[:, :, 0:5], axis=2).ravel())
kl = np.concatenate(kl)
print("   FULL k-pair length (32 MFMAs): p05 %.0f p25 %.0f median %.0f p75 %.0f p95 %.0f" % tuple(np.percentile(kl, [5, 25, 50, 75, 95])))
