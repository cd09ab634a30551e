cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  d=/tmp/pmc_$RANDOM
  rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/microbench_pair.py 16 > /dev/null 2>&1
  python3 - $d <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "pair" if "conv_pair" in r["Kernel_Name"] else ("conv_ws" if "conv_ws" in r["Kernel_Name"] else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(d.items())}, "n=", len(next(iter(d.values()))))
PY
done
