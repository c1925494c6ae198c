#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + PMC passes of the bench command and of four GEMM shapes; everything lands in
# gpurun_out/ (copy what is to be judged into profiles/).   bash scripts/capture_profiles.sh <tag>
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# Single-stream runs for the per-kernel numbers (with the weight-gradient stream on, dW launches overlap the dX chain and the traced
# durations of both stretch over each other); the default (overlapped) step is traced separately below.
export MOREC_WGRAD_STREAM=0
BENCH="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary"
rm -rf /tmp/prof; mkdir -p /tmp/prof
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o kt -- $BENCH > $O/${TAG}_prof_bench_line.json 2> /dev/null
NS=$(python -c "import json,sys; print(json.loads([l for l in open('$O/${TAG}_prof_bench_line.json') if l.startswith('{')][-1])['steps_executed'])")
python $R/scripts/prof_summary.py /tmp/prof/kt_results.db $NS "$TAG: MOREC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary ($NS steps traced: 2 warm-up + 4 timed + 5 of the instrumented pass; single stream)" > $O/${TAG}_bench_kernel_stats.csv
rm -f /tmp/prof/ko_results.db
MOREC_WGRAD_STREAM=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ko -- $BENCH > /dev/null 2>&1
python $R/scripts/prof_summary.py /tmp/prof/ko_results.db $NS "$TAG: default step (weight-gradient stream ON: kernels of the two streams overlap, their durations add up to more than the step)" > $O/${TAG}_bench_kernel_stats_overlap.csv
# the fp16_res32 mode (fp32 residual stream): single-stream kernel trace
rm -f /tmp/prof/kr_results.db
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o kr -- $BENCH --dtype fp16_res32 > /dev/null 2>&1
python $R/scripts/prof_summary.py /tmp/prof/kr_results.db $NS "$TAG fp16_res32: MOREC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- bench.py --dtype fp16_res32 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary ($NS steps traced; single stream)" > $O/${TAG}_res32_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o pf -- $BENCH > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof -o pw -- $BENCH > /dev/null 2>&1
python $R/scripts/pmc_traffic.py /tmp/prof/pf_results.db /tmp/prof/pw_results.db $O/${TAG}_prof_bench_line.json $O/${TAG}_gemm_pmc.json
{
for spec in "51200 2304 768 bias" "51200 3072 768 gelu" "51200 3072 768 dmul" "51200 768 3072 nt" "51200 2304 768 tn" "51200 768 3072 tn"; do
  echo "# rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT -- python scripts/gemm_one.py $spec   (per-dispatch sums over the chip, 3 dispatches averaged)"
  rm -f /tmp/prof/sq_results.db
  timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/prof -o sq -- python $R/scripts/gemm_one.py $spec > /dev/null 2>&1
  python $R/scripts/pmc_summary.py /tmp/prof/sq_results.db "%gemm%8p%"
done
} > $O/${TAG}_gemm_pmc_sq.txt 2>&1
# Swin-T step: kernel trace
rm -f /tmp/prof/ks_results.db
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ks -- python $R/bench.py --tower swin_tiny --batch 64 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > $O/${TAG}_prof_swin_line.json 2> /dev/null
NS2=$(python -c "import json,sys; print(json.loads([l for l in open('$O/${TAG}_prof_swin_line.json') if l.startswith('{')][-1])['steps_executed'])")
python $R/scripts/prof_summary.py /tmp/prof/ks_results.db $NS2 "$TAG swin_tiny B=64 (704 images/step): MOREC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- bench.py --tower swin_tiny --batch 64 --steps 4 --warmup 2 ($NS2 steps traced)" > $O/${TAG}_swin_tiny_kernel_stats.csv
# Swin-T / Swin-B: HBM-side bytes of the GEMM launches (FETCH_SIZE / WRITE_SIZE, separate passes) -> <tag>_swin_{tiny,base}_gemm_pmc.json,
# which bench.py matches by signature for the vision lines' roofline.traffic
for spec in "swin_tiny 64" "swin_base 32"; do
  set -- $spec
  SB="python $R/bench.py --tower $1 --batch $2 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary"
  MOREC_WGRAD_STREAM=0 timeout 200 $SB > $O/${TAG}_prof_$1_line.json 2> /dev/null
  rm -f /tmp/prof/sf_results.db /tmp/prof/swr_results.db
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof -o sf -- $SB > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof -o swr -- $SB > /dev/null 2>&1
  python $R/scripts/pmc_traffic.py /tmp/prof/sf_results.db /tmp/prof/swr_results.db $O/${TAG}_prof_$1_line.json $O/${TAG}_$1_gemm_pmc.json
done
# scoring kernels at the 8-rank pooled column count (Nr = 2560 rows x Nc = 21 504 columns, D = 512): HBM-side bytes per launch, then
# the kernel trace of the same command
{
for c in FETCH_SIZE WRITE_SIZE; do
  echo "# rocprofv3 --kernel-trace --pmc $c -- python scripts/ce_pooled_bench.py 20 0 512 8   (per-dispatch averages, KiB; FETCH_SIZE to be doubled on gfx950: MI355X_MICROARCH.md, HBM section)"
  rm -f /tmp/prof/ce_results.db
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/prof -o ce -- python $R/scripts/ce_pooled_bench.py 20 0 512 8 > /dev/null 2>&1
  python $R/scripts/pmc_summary.py /tmp/prof/ce_results.db "%"  | grep -v "at::native"
done
echo "# algorithmic bytes at this size: forward (Nr + Nc) D 2 + 13 Nc + 8 Nr = 24.9 MB; backward 2 (Nr + Nc) D 2 + 4 Nr + Nr D 2 + Nc D 4 = 95.9 MB (dE handed out in fp32)"
rm -f /tmp/prof/cek_results.db
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o cek -- python $R/scripts/ce_pooled_bench.py 20 0 512 8 > $O/${TAG}_ce_pooled_bench.txt 2>&1
python $R/scripts/prof_summary.py /tmp/prof/cek_results.db 1 "$TAG scoring at the 8-rank pooled size: rocprofv3 --kernel-trace --stats -- python scripts/ce_pooled_bench.py 20 0 512 8 (2 + 21 forward calls, 1 + 20 backward calls)"
python $R/scripts/ce_pooled_bench.py 20 0 512
python $R/scripts/ce_pooled_bench.py 20 1 512 8
} > $O/${TAG}_scoring_pmc.txt 2>&1

# round 5: tile height of gemm8p per shape (256-row tiles vs automatic vs forced 224 / 192) and the 64 x 64 attention tile against the VALU fallback
{
echo "# python scripts/tmr_bench.py  -- per launch: <gemm8p_tmr mode>:<time>/<TFLOP/s>; modes 0 = 256-row tiles, 1 = automatic (pick_tmr), 224 / 192 forced; 20 launches each, same process"
python $R/scripts/tmr_bench.py
} > $O/${TAG}_tile_height.txt 2>&1
{
echo "# python scripts/attn64_bench.py (T = 50, 2688 titles, 12 heads x 64, fp16, dropout 0.1): attention_mfma64.hip, then MOREC_ATTN_MFMA64=0 (exact-fp32 VALU kernels)"
python $R/scripts/attn64_bench.py
MOREC_ATTN_MFMA64=0 python $R/scripts/attn64_bench.py
} > $O/${TAG}_attn64.txt 2>&1
# ID tower kernel trace (two streams, as it runs)
rm -f /tmp/prof/ki_results.db
MOREC_WGRAD_STREAM=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ki -- python $R/bench.py --tower id --batch 128 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/scripts/prof_summary.py /tmp/prof/ki_results.db 34 "$TAG id tower: rocprofv3 --kernel-trace --stats -- bench.py --tower id --batch 128 --steps 20 --warmup 5 (34 steps traced, two streams)" > $O/${TAG}_id_tower_kernel_stats.csv
