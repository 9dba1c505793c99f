set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_nl3
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || true
grep -i -o "SQC\?_[A-Z_]*ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_VALU\|TCP_[A-Z_]*STALL[A-Z_]*" $OUT/avail.txt | sort -u | head -40 > $OUT/avail_short.txt
run() { name=$1; shift; rocprofv3 "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/time_nltv.py > $OUT/$name.log 2>&1; }
run sq1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
run sq3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq4 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
for path in glob.glob("gpurun_out/prof_nl3/**/*_counter_collection.csv",recursive=True):
    for row in csv.DictReader(open(path)):
        n=row["Kernel_Name"]
        if "nl_fused" in n or "nl_dual" in n or "nl_primal" in n:
            k=n.split("(")[0][-40:]
            s=acc[k][row["Counter_Name"]]; s[0]+=float(row["Counter_Value"]); s[1]+=1
for k,v in acc.items():
    print(k)
    for c,(t,n) in sorted(v.items()): print("   %-24s %14.1f  (n=%d)"%(c,t/n,n))
PY
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 $OUT/fetch $OUT/write
