mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r02g/counters.txt 2>&1
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/r02g/$n -- python3 tools/tree_roofline.py --games 65536 --steps 30 --preroll 1200 > gpurun_out/r02g/$n.out 2>&1
}
run a TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run b TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run c TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
run d TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_PENDING_STALL_CYCLES_sum
run e TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run f FETCH_SIZE
run g WRITE_SIZE
run h SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
python - <<'PY'
import csv,glob,collections,json
out={}
for d in "abcdefgh":
    fs=glob.glob(f'gpurun_out/r02g/{d}/**/*counter_collection.csv',recursive=True)
    if not fs: out[d]='none'; continue
    acc=collections.defaultdict(float); seen=set()
    for row in csv.DictReader(open(fs[0])):
        if 'c4_step_kernel' not in row['Kernel_Name']: continue
        acc[row['Counter_Name']]+=float(row['Counter_Value']); seen.add(row['Dispatch_Id'])
    out[d]={k:v/max(1,len(seen)) for k,v in acc.items()}; out[d]['dispatches']=len(seen)
json.dump(out,open('gpurun_out/r02g/summary.json','w'),indent=1)
print(json.dumps(out,indent=1))
PY
