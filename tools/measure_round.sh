set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export TMPDIR=/tmp
# 1. kernel trace + device busy of the default bench (3 steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r07b -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/r07b_bench_under_rocprof.json 2> $O/r07b.log || true
python tools/summarize_prof.py $O/prof_r07b $O/r07b_kernel_trace 5 > $O/r07b_kernel_trace_summary.txt 2>&1 || true
rm -rf $O/prof_r07b
grep "device busy" $O/r07b_kernel_trace_summary.txt || true
# 2. two ranks sharing the GPU
PF_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/r07_bench_2ranks_shared_gpu.json 2> $O/r07_bench_2ranks.log || true
tail -c 600 $O/r07_bench_2ranks_shared_gpu.json
# 3. CLI wall at 5 M: default (one process) and --detach-teardown
W=$(mktemp -d /tmp/pf_t.XXXX)
python tools/make_graph.py $W/in 5000000 1000 | tail -1
cd $W
for mode in "" "--detach-teardown"; do for i in 1 2; do
t0=$(date +%s%N)
$R/ploidyfrost_amd/csrc/ploidyfrost -g $W/in/g.gfa -d $W/in/g_kmc -o x -l 5 -u 1000 -t 32 $mode > log.txt
echo "cli 5M [$mode] wall $(( ($(date +%s%N) - t0) / 1000000 )) ms; $(grep -E 'findSuperBubble\(\):  Real|PloidyEstimation\(\):  Real' log.txt | tr '\n' ' ')"
sleep 1
done; done
cd $R; rm -rf $W
