#!/bin/bash
# Round-3 measurement batch (run on the GPU box from the repo root): default bench line, then the four workloads under
# rocprofv3 --kernel-trace --stats.  Every GPU step has its own time limit; the script stops at the first failure.
set -e
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats --output-format csv"
timeout -k 10 600 python bench.py > gpurun_out/r3/bench_bf16_b512_final.json 2> gpurun_out/r3/bench_final.err
echo "bench done"
timeout -k 10 300 $P -d gpurun_out/r3/prof_bf16 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32 > gpurun_out/r3/bench_bf16_b512_under_rocprof.json 2> gpurun_out/r3/prof_bf16.err
echo "bf16 profiled"
timeout -k 10 300 $P -d gpurun_out/r3/prof_fp32 -- python bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_fp32_b512_under_rocprof.json 2> gpurun_out/r3/prof_fp32.err
echo "fp32 profiled"
timeout -k 10 300 $P -d gpurun_out/r3/prof_train -- python bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_train_under_rocprof.json 2> gpurun_out/r3/prof_train.err
echo "train profiled"
timeout -k 10 300 $P -d gpurun_out/r3/prof_vqvae -- python bench.py --workload vqvae --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3/bench_vqvae_under_rocprof.json 2> gpurun_out/r3/prof_vqvae.err
echo "vqvae profiled"
timeout -k 10 300 python bench.py --workload train --no-cpu-baseline > gpurun_out/r3/bench_train_b512.json 2> gpurun_out/r3/bench_train.err
timeout -k 10 300 python bench.py --workload vqvae > gpurun_out/r3/bench_vqvae_b256_t88.json 2> gpurun_out/r3/bench_vqvae.err
for w in bf16 fp32 train vqvae; do
  f=$(find gpurun_out/r3/prof_$w -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/r3/${w}_kernel_stats.csv
  rm -rf gpurun_out/r3/prof_$w
done
echo "all done"
