#!/bin/bash
# Round-5 measurements on one MI355X (run from the repo root on the GPU box: `gpurun -- tools/gpu_round5.sh <sections>`).
# Output goes to gpurun_out/r05/; what is to be judged is copied into profiles/ afterwards.
#   micro     tools/microbench/ldsdma_pass: LDS-DMA / s_setprio / spread-issue variants of a pass-shaped persistent tile loop
#   chunk     same-box A/B of the chunked key switch (SEALHIP_KS_CHUNK / SEALHIP_KS_LANES) on the headline, rotate_c5, bfv_c4
#   prio      same-box A/B of the s_setprio variants of the real kernels (seal_amd/lib/variants/prio1.so, prio2.so)
#   c2        configs[1] chain: default vs -DSEALHIP_KS_NT=0 (variants/nt0.so), 5 processes each
#   newtests  the GPU tests added this round
#   bfvpmc    counter pass over the bfv_c4 workload (behz_*, ntt2_*<7,0>, ks2<7,0>)
#   chunk2 chunktrace pipe mall nttvar wg8k pack packtrace p1bound lean1 pbbound fchunks invwant t2wg bfvtrace2 multi fuzzchunk
#             the other A/Bs and traces of the round, each behind the profiles/r05_* file that quotes it
#   tests     pytest -m gpu + smoke
#   bench     python bench.py (default line)
#   trace     rocprofv3 --kernel-trace --stats of a short bench + the step's time line
# Library variants the A/B sections name (seal_amd/lib/variants/NAME.so; build first with tools/quick/build_variant.sh NAME "FLAGS"):
#   prio1 / prio2   -DSEALHIP_PRIO=1 / =2            nt0      -DSEALHIP_KS_NT=0           p2w2 / p1w2  -DSEALHIP_FP_WAVES_P2=2 / _P1=2
#   wg2k/8k/16k     -DSEALHIP_NTT_WG_TARGET=2048 ... (8192 is the default since)          nopack   -DSEALHIP_MID_PACK=0
#   packw4          -DSEALHIP_PACK_WAVES_P1=4        nolean1  -DSEALHIP_P1_PLAIN_LEAN=0 (the default since; =1 is the variant now)
#   p1noload / p1nostore / p1nomem   -DSEALHIP_P1_NOLOAD / -DSEALHIP_P1_NOSTORE / both      t2wg2k/8k/16k  -DSEALHIP_TAIL2_WG_TARGET=...
#   ab              "" (development switches only: SEALHIP_AB_SKIP_INV_PB for pbbound)      invw2k   (removed: the inverse's rule is in the launcher)
# Sections whose experiment code was removed again (pipe: SEALHIP_KS_PIPE; mall: SEALHIP_NTT_CHUNK_MIB) are kept as the record of the
# command line behind profiles/r05_ks_chunked.txt / r05_ntt_mall_chunks.txt.
set -u
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r05; mkdir -p $O
export SEALHIP_ABORT_TRACE=$O/abort_trace.txt
common="--no-cpu-baseline --no-pmc --no-verify --no-children"
[ $# -eq 0 ] && set -- micro chunk
for what in "$@"; do
  echo "=== $what $(date +%T)"
  case $what in
  micro)
    timeout 300 tools/microbench/ldsdma_pass 4 > $O/ldsdma_pass.txt 2>&1; tail -80 $O/ldsdma_pass.txt ;;
  chunk)
    tools/ab.sh --rounds ${ROUNDS:-2} --out gpurun_out/r05/ab_chunk \
      off:default:SEALHIP_KS_CHUNK=0 c32x2:default c32x1:default:SEALHIP_KS_LANES=1 c32x3:default:SEALHIP_KS_LANES=3 \
      c64x2:default:SEALHIP_KS_CHUNK=64 c16x4:default:SEALHIP_KS_CHUNK=16,SEALHIP_KS_LANES=4 2>&1 | tee $O/ab_chunk.txt
    tools/ab.sh --rounds 1 --workload rotate_c5 --out gpurun_out/r05/ab_chunk_rot off:default:SEALHIP_KS_CHUNK=0 c8x2:default:SEALHIP_KS_CHUNK=8 c16x2:default:SEALHIP_KS_CHUNK=16 2>&1 | tee $O/ab_chunk_rot.txt
    tools/ab.sh --rounds 1 --workload bfv_c4 --out gpurun_out/r05/ab_chunk_bfv off:default:SEALHIP_KS_CHUNK=0 auto:default auto3:default:SEALHIP_KS_LANES=3 2>&1 | tee $O/ab_chunk_bfv.txt ;;
  chunk2)
    tools/ab.sh --rounds ${ROUNDS:-3} --out gpurun_out/r05/ab_chunk2 off:default:SEALHIP_KS_CHUNK=0 c32x3:default c64x2:default:SEALHIP_KS_CHUNK=64,SEALHIP_KS_LANES=2 c32x1:default:SEALHIP_KS_LANES=1 2>&1 | tee $O/ab_chunk2.txt ;;
  mall)
    tools/ab.sh --rounds ${ROUNDS:-2} --workload ntt --out gpurun_out/r05/ab_mall base:default m64x2:default:SEALHIP_NTT_CHUNK_MIB=64 m96x2:default:SEALHIP_NTT_CHUNK_MIB=96 \
      m128x2:default:SEALHIP_NTT_CHUNK_MIB=128 m96x3:default:SEALHIP_NTT_CHUNK_MIB=96,SEALHIP_NTT_LANES=3 m64x3:default:SEALHIP_NTT_CHUNK_MIB=64,SEALHIP_NTT_LANES=3 \
      m256x2:default:SEALHIP_NTT_CHUNK_MIB=256 m96x1:default:SEALHIP_NTT_CHUNK_MIB=96,SEALHIP_NTT_LANES=1 2>&1 | tee $O/ab_mall.txt ;;
  bfvtrace)
    bash tools/quick/timeline_other.sh > $O/timeline_other.log 2>&1; cp gpurun_out/tl_rot/*.txt $O/ 2>/dev/null; tail -60 $O/timeline_other.log ;;
  bfvtrace2)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/prof_bfv -o t -- python $REPO/bench.py --workload bfv_c4 --steps 4 --warmup 1 $common > $O/prof_bfv.log 2>&1)
    DB=$(find $O/prof_bfv -name "*.db" | head -1)
    python tools/step_timeline.py $DB --anchor behz_floor_sk --step -3 > $O/timeline_bfv_step.txt 2>&1; rm -rf $O/prof_bfv; tail -70 $O/timeline_bfv_step.txt ;;
  nttvar)
    tools/ab.sh --rounds ${ROUNDS:-2} --workload ntt --out gpurun_out/r05/ab_nttvar base:default p2w2h3:p2w2:SEALHIP_P2_HOIST=3 p2w2h4:p2w2 p1w2:p1w2 wg2k:wg2k wg8k:wg8k wg16k:wg16k 2>&1 | tee $O/ab_nttvar.txt ;;
  multi)
    (timeout 1500 python -m pytest tests/test_gpu_multi.py -q -x -rs > $O/pytest_multi.txt 2>&1; echo "rc=$?" >> $O/pytest_multi.txt); tail -15 $O/pytest_multi.txt ;;
  pipe)
    tools/ab.sh --rounds ${ROUNDS:-2} --out gpurun_out/r05/ab_pipe off:default:SEALHIP_KS_CHUNK=0 c32x3:default p32r2:default:SEALHIP_KS_PIPE=1,SEALHIP_KS_LANES=2 p32r3:default:SEALHIP_KS_PIPE=1 \
      p16r3:default:SEALHIP_KS_PIPE=1,SEALHIP_KS_CHUNK=16 p16r4:default:SEALHIP_KS_PIPE=1,SEALHIP_KS_CHUNK=16,SEALHIP_KS_LANES=4 p64r2:default:SEALHIP_KS_PIPE=1,SEALHIP_KS_CHUNK=64,SEALHIP_KS_LANES=2 2>&1 | tee $O/ab_pipe.txt
    SEALHIP_KS_PIPE=1 tools/ab.sh --rounds 1 --trace --out gpurun_out/r05/ab_pipe_trace p32r3:default:SEALHIP_KS_PIPE=1 2>&1 | tail -45 | tee $O/ab_pipe_trace.txt
    (SEALHIP_KS_PIPE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ks_chunked or batch256" > $O/pytest_pipe.txt 2>&1; echo "rc=$?" >> $O/pytest_pipe.txt); tail -3 $O/pytest_pipe.txt ;;
  pbbound)
    tools/ab.sh --rounds ${ROUNDS:-3} --out gpurun_out/r05/ab_pbbound base:ab nopb:ab:SEALHIP_AB_SKIP_INV_PB=38 nopb0:ab:SEALHIP_AB_SKIP_INV_PB=0 2>&1 | tee $O/ab_pbbound.txt ;;
  wg8k)
    tools/ab.sh --rounds ${ROUNDS:-4} --workload ntt --out gpurun_out/r05/ab_wg8k base:default wg8k:wg8k 2>&1 | tee $O/ab_wg8k.txt
    tools/ab.sh --rounds 2 --out gpurun_out/r05/ab_wg8k_head base:default wg8k:wg8k 2>&1 | tee $O/ab_wg8k_head.txt
    tools/ab.sh --rounds 2 --workload bfv_c4 --out gpurun_out/r05/ab_wg8k_bfv base:default wg8k:wg8k 2>&1 | tee $O/ab_wg8k_bfv.txt ;;
  pack)
    (timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_ntt" > $O/pytest_pack.txt 2>&1; echo "rc=$?" >> $O/pytest_pack.txt); tail -4 $O/pytest_pack.txt
    tools/ab.sh --rounds ${ROUNDS:-3} --workload ntt --out gpurun_out/r05/ab_pack nopack:nopack pack:default packw4:packw4 2>&1 | tee $O/ab_pack.txt ;;
  packtrace)
    for v in default; do
      if [ $v = default ]; then cp seal_amd/lib/libsealhip.so /tmp/keep.so; else cp seal_amd/lib/libsealhip.so /tmp/keep.so; cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so; fi
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o t -- python $REPO/bench.py --ntt-only $common > $O/prof_$v.log 2>&1)
      DB=$(find $O/prof_$v -name "*.db" | head -1); python tools/rocpd_summary.py $DB | grep -E 'kernel|ntt2_fwd' | head -8; rm -rf $O/prof_$v
      cp /tmp/keep.so seal_amd/lib/libsealhip.so
    done ;;
  fuzzchunk)
    # random operation sequences, deferred tails and the soak with every key switch forced into chunks of ONE item on three lanes
    (SEALHIP_KS_SPLIT=1 SEALHIP_KS_CHUNK=1 SEALHIP_KS_LANES=3 timeout 1200 python -m pytest tests/test_fuzz.py tests/test_soak.py tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or sequences or soak or pipeline or deferred or north_star_batch16" > $O/pytest_fuzzchunk.txt 2>&1; echo "rc=$?" >> $O/pytest_fuzzchunk.txt); tail -5 $O/pytest_fuzzchunk.txt ;;
  p1bound)
    # per-kernel averages of pass 1 with its loads / stores / both removed (timing only: wrong words)
    for v in default p1noload p1nostore p1nomem; do
      cp seal_amd/lib/libsealhip.so /tmp/keep.so; [ $v = default ] || cp seal_amd/lib/variants/$v.so seal_amd/lib/libsealhip.so
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o t -- python $REPO/bench.py --ntt-only $common > $O/prof_$v.log 2>&1)
      DB=$(find $O/prof_$v -name "*.db" | head -1); echo "== $v: $(tail -1 $O/prof_$v.log | python -c 'import json,sys; j=json.loads(sys.stdin.read())["roofline"]; print(j["achieved"], "GB/s", j["ms_per_launch"], "ms per launch")')"; python tools/rocpd_summary.py $DB | grep -E 'ntt2_fwd_p' | head -4; rm -rf $O/prof_$v
      cp /tmp/keep.so seal_amd/lib/libsealhip.so
    done ;;
  lean1)
    (timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "test_ntt or north_star_config or bfv_pipeline" > $O/pytest_lean1.txt 2>&1; echo "rc=$?" >> $O/pytest_lean1.txt); tail -3 $O/pytest_lean1.txt
    tools/ab.sh --rounds ${ROUNDS:-3} --workload ntt --out gpurun_out/r05/ab_lean1 before:nolean1 lean:default 2>&1 | tee $O/ab_lean1.txt
    tools/ab.sh --rounds 2 --out gpurun_out/r05/ab_lean1_head before:nolean1 lean:default 2>&1 | tee $O/ab_lean1_head.txt ;;
  fchunks)
    tools/ab.sh --rounds 2 --workload c2 --out gpurun_out/r05/ab_fchunks def:default f256:default:SEALHIP_NTT_FCHUNKS=256 f512:default:SEALHIP_NTT_FCHUNKS=512 f2048:default:SEALHIP_NTT_FCHUNKS=2048 f4096:default:SEALHIP_NTT_FCHUNKS=4096 2>&1 | tee $O/ab_fchunks.txt ;;
  invwant)
    tools/ab.sh --rounds 3 --workload c2 --out gpurun_out/r05/ab_invwant before:invw2k after:default 2>&1 | tee $O/ab_invwant.txt
    (timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "test_ntt" > $O/pytest_invwant.txt 2>&1; echo "rc=$?" >> $O/pytest_invwant.txt); tail -3 $O/pytest_invwant.txt ;;
  t2wg)
    tools/ab.sh --rounds 3 --out gpurun_out/r05/ab_t2wg base:default t2wg2k:t2wg2k t2wg8k:t2wg8k t2wg16k:t2wg16k 2>&1 | tee $O/ab_t2wg.txt
    tools/ab.sh --rounds 2 --workload rotate_c5 --out gpurun_out/r05/ab_t2wg_rot base:default t2wg8k:t2wg8k t2wg16k:t2wg16k 2>&1 | tee $O/ab_t2wg_rot.txt ;;
  wide)
    (timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "north_star or batch1024 or ckks_pipeline" > $O/pytest_wide.txt 2>&1; echo "rc=$?" >> $O/pytest_wide.txt); tail -3 $O/pytest_wide.txt
    tools/ab.sh --rounds ${ROUNDS:-3} --trace --out gpurun_out/r05/ab_wide before:nowide wide:default 2>&1 | grep -E 'round|ckks_multiply|== ' | tee $O/ab_wide.txt ;;
  ewgrid)
    tools/ab.sh --rounds 2 --workload bfv_c4 --out gpurun_out/r05/ab_ew_bfv base:default ewfull:ewfull ew16k:ew16k 2>&1 | tee $O/ab_ew_bfv.txt
    tools/ab.sh --rounds 2 --workload rotate_c5 --out gpurun_out/r05/ab_ew_rot base:default ewfull:ewfull ew16k:ew16k 2>&1 | tee $O/ab_ew_rot.txt
    tools/ab.sh --rounds 2 --out gpurun_out/r05/ab_ew_head base:default ewfull:ewfull ew16k:ew16k 2>&1 | tee $O/ab_ew_head.txt ;;
  chunktrace)
    tools/ab.sh --rounds 1 --trace --out gpurun_out/r05/ab_chunk_trace c32x2:default 2>&1 | tee $O/ab_chunk_trace.txt ;;
  prio)
    tools/ab.sh --rounds ${ROUNDS:-2} --out gpurun_out/r05/ab_prio --check "north_star_config or test_ntt" base:default prio1:prio1 prio2:prio2 2>&1 | tee $O/ab_prio.txt
    tools/ab.sh --rounds ${ROUNDS:-2} --workload ntt --out gpurun_out/r05/ab_prio_ntt base:default prio1:prio1 prio2:prio2 2>&1 | tee $O/ab_prio_ntt.txt ;;
  c2)
    tools/ab.sh --rounds 5 --workload c2 --out gpurun_out/r05/ab_c2 nt31:default nt0:nt0 2>&1 | tee $O/ab_c2.txt ;;
  newtests)
    (timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "ks_chunked or batch512 or batch256 or batch1024" > $O/pytest_new.txt 2>&1; echo "rc=$?" >> $O/pytest_new.txt); tail -5 $O/pytest_new.txt ;;
  bfvpmc)
    timeout 900 python tools/pmc_table.py --bench-args "--workload bfv_c4 --steps 2 --warmup 1 $common" --filter "" --groups 0,1,7,8 > $O/bfv_c4_counters.txt 2>&1; tail -120 $O/bfv_c4_counters.txt ;;
  tests)
    (timeout 1800 python -m pytest tests -m gpu -q -rs > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt); tail -6 $O/pytest.txt
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt); tail -1 $O/smoke.txt ;;
  bench)
    timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json ;;
  trace)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 $common > $O/prof.log 2>&1)
    DB=$(find $O/prof -name "*.db" | head -1)
    python tools/rocpd_summary.py $DB > $O/rocprof_bench_kernel_stats.txt; python tools/step_timeline.py $DB > $O/step_timeline.txt
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $O/rocprofv3_stats_kernel_stats.csv; rm -rf $O/prof
    tail -40 $O/step_timeline.txt ;;
  esac
done
[ -s $O/abort_trace.txt ] && { echo "ABORT TRACE:"; cat $O/abort_trace.txt; }
exit 0
