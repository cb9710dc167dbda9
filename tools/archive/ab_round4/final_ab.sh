#!/bin/bash
# Same-box A/B of round 4's switches (one gpurun call): writes gpurun_out/r4_ab_switches.jsonl, one bench line per arm.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4_ab_switches.jsonl; : > $O
run() { tag=$1; shift; env "$@" python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = {n.split(' (')[0].split('/')[0]: {'ms': round(v['ms_total'], 1), 'rate': round(v.get('TFLOP/s', v.get('GB/s', 0)))} for n, v in d.get('kernels', {}).items()}
print(json.dumps({'arm': '$tag', 'config': '$CFG', 'frames_per_s': round(d['value'], 2), 'ms_per_video': round(d['ms_per_step'], 1), 'dit_step_ms': round(d.get('dit_step_ms', 0), 2), 'vae_decode_ms': round(d.get('vae_decode_ms', 0), 1), 'kernels': k}))" >> $O; }
CFG=c2
run default A=1
run "LTX_Q2_FOLD=0 (stand-alone cross-attention q-norm)" LTX_Q2_FOLD=0
run "LTX_NORM_PRESUM=0 (row-reducing RMS norms)" LTX_NORM_PRESUM=0
run "LTX_NORM_LEAN=0 (general presum map)" LTX_NORM_LEAN=0
run "LTX_GEMM_RING=0 LTX_GEMM_SPLIT_RING=0 (small-M layers on gemm_big)" LTX_GEMM_RING=0 LTX_GEMM_SPLIT_RING=0
run "LTX_CONV_OUT_HALO=0 (conv_out on the per-tap tile)" LTX_CONV_OUT_HALO=0
run "all of round 4 off" LTX_Q2_FOLD=0 LTX_NORM_PRESUM=0 LTX_NORM_LEAN=0 LTX_GEMM_RING=0 LTX_GEMM_SPLIT_RING=0 LTX_CONV_OUT_HALO=0
run default-again A=1
CFG=c1
run default A=1
run "LTX_GEMM_RING_SPEC=0 (every wave loads and multiplies)" LTX_GEMM_RING_SPEC=0
run "LTX_GEMM_SPLIT_RING=0 (the round-3 split rule, ring tiles)" LTX_GEMM_SPLIT_RING=0
run "LTX_GEMM_RING=0 LTX_GEMM_SPLIT_RING=0 (round 3: gemm_big)" LTX_GEMM_RING=0 LTX_GEMM_SPLIT_RING=0
run default-again A=1
CFG=c4
run default A=1
run "LTX_GEMM_BIG_CONV_MINM=256 (edge-tile convs on the 128 x 128 kernel)" LTX_GEMM_BIG_CONV_MINM=256
CFG=c5
run default A=1
run "LTX_QKNORM_FUSED8=0" LTX_QKNORM_FUSED8=0
cut -c1-200 $O
