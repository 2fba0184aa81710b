"""Aggregate a rocprofv3 stochastic PC-sampling run (csv [+ json]) into per-kernel tables: issued / not-issued by reason, by instruction
type, and the top instructions by samples.  usage: python tools/pcsamp_summary.py <dir> [kernel-regex]
Written defensively (the CSV columns of the beta tool are discovered, not assumed)."""
import csv, glob, os, re, sys, collections, json

d = sys.argv[1]
kre = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_contract16|k_corr|k_sweep_small|k_acq")
csv.field_size_limit(1 << 30)


def find(pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


kt = find("*kernel_trace.csv")
disp = {}
if kt:
    with open(kt) as f:
        for row in csv.DictReader(f):
            did = row.get("Dispatch_Id") or row.get("Dispatch_ID")
            disp[did] = (row.get("Kernel_Name", "?"), int(row.get("End_Timestamp", 0)) - int(row.get("Start_Timestamp", 0)))
print("# dispatches in kernel trace:", len(disp))
ps = find("*pc_sampling_stochastic.csv") or find("*pc_sampling*.csv")
if not ps:
    print("no pc sampling csv found under", d)
    sys.exit(0)
with open(ps) as f:
    rd = csv.DictReader(f)
    cols = rd.fieldnames
    print("# columns:", cols)
    per = collections.defaultdict(lambda: collections.Counter())
    tot = collections.Counter()
    for row in rd:
        kn = disp.get(row.get("Dispatch_Id"), ("?", 0))[0]
        short = re.sub(r"\(.*", "", kn)[:60]
        if not kre.search(kn):
            tot[("other", short)] += 1
            continue
        tot[("kept", short)] += 1
        issued = row.get("Wave_Issued_Instruction", "?")
        ity = row.get("Instruction_Type", "?")
        why = row.get("Stall_Reason", "?")
        ins = row.get("Instruction", "?")
        per[short][("A", issued, why)] += 1
        per[short][("B", ity, issued)] += 1
        per[short][("C", ins, issued, why)] += 1
        per[short][("D", row.get("Wave_Count", "?"))] += 1
print("# samples per kernel:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("   ", k, v)
for kn, c in per.items():
    n = sum(v for k, v in c.items() if k[0] == "A")
    print("\n==== %s: %d samples" % (kn, n))
    print("-- issued / reason-not-issued")
    for k, v in sorted(((k, v) for k, v in c.items() if k[0] == "A"), key=lambda kv: -kv[1]):
        print("   %6.2f %%  issued=%s reason=%s" % (100.0 * v / n, k[1], k[2]))
    print("-- instruction type x issued")
    for k, v in sorted(((k, v) for k, v in c.items() if k[0] == "B"), key=lambda kv: -kv[1]):
        print("   %6.2f %%  type=%s issued=%s" % (100.0 * v / n, k[1], k[2]))
    print("-- waves resident on the SIMD at the sample")
    for k, v in sorted(((k, v) for k, v in c.items() if k[0] == "D"), key=lambda kv: -kv[1]):
        print("   %6.2f %%  wave_count=%s" % (100.0 * v / n, k[1]))
    print("-- top 60 (instruction, issued, reason)")
    for k, v in sorted(((k, v) for k, v in c.items() if k[0] == "C"), key=lambda kv: -kv[1])[:60]:
        print("   %6.2f %%  %-70s issued=%s reason=%s" % (100.0 * v / n, k[1][:70], k[2], k[3]))
js = find("*results.json")
if js and os.path.getsize(js) < (1 << 31):
    try:
        with open(js) as f:
            J = json.load(f)
        root = J["rocprofiler-sdk-tool"][0]
        print("\n# json top-level keys:", list(root.keys()))
        br = root.get("buffer_records", {})
        print("# buffer_records keys:", {k: len(v) if hasattr(v, "__len__") else v for k, v in br.items()})
        for k, v in br.items():
            if "pc_sampl" in k and len(v):
                print("# first %s record:" % k, json.dumps(v[0])[:1500])
    except Exception as e:  # noqa
        print("json summary failed:", repr(e))
