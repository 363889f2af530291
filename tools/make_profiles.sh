#!/bin/bash
# Regenerates the measurement artifacts of one round on the GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh r01      -> gpurun_out/<tag>_*  (copy the ones to keep into profiles/)
# Separate rocprofv3 passes: kernel trace, then one PMC counter per pass (never combined with other trace domains).
# The script is long (the fuzz sweeps alone take minutes): `tools/make_profiles.sh r05 A` runs the first half (bench lines, rocprofv3
# passes, configs 3-5), `... r05 B` the second (mid-size kernel, API paths, solves, sweeps), no second argument both.
# The per-kernel passes (kernel trace, PMC) run `BAYENS_NO_LOOKAHEAD=1 bench.py`: kernels strictly one after the other - the setting
# bench.py's own roofline block measures in (the timed headline step runs the look-ahead stream beside the main one, on another
# hardware queue: two kernels at a time, whose recorded durations include the time they share the device).  The trace of the plain
# command is kept beside it (<tag>_kernel_stats_overlapped.txt).
set -u
TAG=${1:-r01}
PART=${2:-AB}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
if [[ $PART == *A* ]]; then
timeout -k 10 500 python3 bench.py > $OUT/${TAG}_bench_final.json 2> $OUT/${TAG}_bench_final.err || exit 1       # the plain command the driver runs
BAYENS_PRINT_STAGES=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_stage_profile.txt || exit 1
BAYENS_PRINT_SHAPES=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > /dev/null 2> $OUT/${TAG}_gemm_shapes_final.txt || exit 1
timeout -k 10 300 python3 bench.py --config c5 > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err || exit 1
timeout -k 10 300 python3 tools/small_phases.py > $OUT/${TAG}_small_phases.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
export BAYENS_NO_LOOKAHEAD=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_kt.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pf -o f -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/${TAG}_pf.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pw -o w -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/${TAG}_pw.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $OUT/${TAG}_pm -o m -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/${TAG}_pm.log 2>&1 || exit 5
unset BAYENS_NO_LOOKAHEAD
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kto -o kto -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_kto.log 2>&1 || exit 2
cd $R
python3 tools/rocpd_stats.py $(find $OUT/${TAG}_kto -name "*.db" | head -1) $OUT/${TAG}_kernel_stats_overlapped.txt > /dev/null
rm -rf $OUT/${TAG}_kto
python3 tools/rocpd_mfma_util.py $(find $OUT/${TAG}_pm -name "*.db" | head -1) $OUT/${TAG}_pmc_mfma_util.txt > /dev/null
rm -rf $OUT/${TAG}_pm
KT=$(find $OUT/${TAG}_kt -name "*.db" | head -1); PF=$(find $OUT/${TAG}_pf -name "*.db" | head -1); PW=$(find $OUT/${TAG}_pw -name "*.db" | head -1)
python3 tools/rocpd_stats.py $KT $OUT/${TAG}_kernel_stats_final.txt > /dev/null
python3 tools/rocpd_pmc.py $PF FETCH_SIZE $OUT/${TAG}_pmc_fetch_size.txt > /dev/null
python3 tools/rocpd_pmc.py $PW WRITE_SIZE $OUT/${TAG}_pmc_write_size.txt > /dev/null
python3 tools/pmc_traffic.py $PF $PW $OUT/${TAG}_pmc_traffic.json > /dev/null
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pf $OUT/${TAG}_pw
# config 5 (the fused small-member kernel): kernel trace of the same command as the bench line
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt5 -o kt5 -- python3 $R/bench.py --config c5 --no-cpu-baseline > $OUT/${TAG}_kt5.log 2>&1 || exit 6
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $OUT/${TAG}_pm5 -o m -- python3 $R/bench.py --config c5 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/${TAG}_pm5.log 2>&1 || exit 19
cd $R
python3 tools/rocpd_stats.py $(find $OUT/${TAG}_kt5 -name "*.db" | head -1) $OUT/${TAG}_kernel_stats_c5.txt > /dev/null
python3 tools/rocpd_mfma_util.py $(find $OUT/${TAG}_pm5 -name "*.db" | head -1) $OUT/${TAG}_pmc_mfma_util_c5.txt > /dev/null
rm -rf $OUT/${TAG}_kt5 $OUT/${TAG}_pm5
timeout -k 10 300 python3 bench.py --config c5 --dtype f32 --grid-points 432 --no-cpu-baseline > $OUT/${TAG}_bench_c5_f32.json 2> $OUT/${TAG}_bench_c5_f32.err || exit 7
timeout -k 10 300 python3 bench.py --config c3 --no-extras > $OUT/${TAG}_bench_c3.json 2> /dev/null || exit 12
timeout -k 10 300 python3 bench.py --config c4 --no-extras > $OUT/${TAG}_bench_c4.json 2> /dev/null || exit 13
timeout -k 10 120 python3 tools/loglik_rows_time.py > $OUT/${TAG}_loglik_rows.txt 2>&1 || exit 8
BAYENS_W2_TRACE=1 timeout -k 10 300 python3 tools/w2_time.py > $OUT/${TAG}_w2_single.txt 2>&1 || exit 9
timeout -k 10 120 python3 tools/diag16_bench.py > $OUT/${TAG}_diag16_bench.txt 2>&1 || exit 10
# one 32 x 2048 Cholesky kernel by kernel (the chain analysis of DESIGN section 9)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_ktp -o ktp -- python3 $R/tools/potrf_trace.py > $OUT/${TAG}_ktp.log 2>&1 || exit 11
cd $R
python3 tools/rocpd_seq.py $(find $OUT/${TAG}_ktp -name "*.db" | head -1) > $OUT/${TAG}_potrf_sequence.txt
rm -rf $OUT/${TAG}_ktp
fi
if [[ $PART == *B* ]]; then
timeout -k 10 300 python3 tools/paper_script.py > $OUT/${TAG}_paper_script.json 2> $OUT/${TAG}_paper_script.err || exit 30
(timeout -k 10 300 python3 tools/monthly_field.py float64; timeout -k 10 300 python3 tools/monthly_field.py float32) > $OUT/${TAG}_monthly_field.json 2> /dev/null || exit 31
timeout -k 10 300 python3 tools/determinism_check.py > $OUT/${TAG}_determinism.txt 2>&1 || exit 32
# the fused mid-size kernel (csrc/medium.hip) at the reference's workload: timing table against the general path, phase stamps,
# kernel trace and matrix-pipe utilisation of one collection fit
(timeout -k 10 300 python3 tools/medium_check.py | grep timing; BAYENS_NO_MEDIUM=1 timeout -k 10 300 python3 tools/medium_check.py | grep timing; timeout -k 10 100 python3 tools/medium_phases.py 165 20 | tail -15) > $OUT/${TAG}_medium_kernel.txt 2>&1 || exit 14
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_ktm -o ktm -- python3 $R/tools/n165_fit.py > $OUT/${TAG}_ktm.log 2>&1 || exit 15
timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $OUT/${TAG}_pmm -o m -- python3 $R/tools/n165_fit.py > $OUT/${TAG}_pmm.log 2>&1 || exit 16
cd $R
python3 tools/rocpd_stats.py $(find $OUT/${TAG}_ktm -name "*.db" | head -1) $OUT/${TAG}_kernel_stats_n165.txt > /dev/null
python3 tools/rocpd_mfma_util.py $(find $OUT/${TAG}_pmm -name "*.db" | head -1) $OUT/${TAG}_pmc_mfma_util_n165.txt > /dev/null
rm -rf $OUT/${TAG}_ktm $OUT/${TAG}_pmm
timeout -k 10 300 python3 tools/api_path_c5.py > $OUT/${TAG}_api_path_c5.json 2> /dev/null || exit 17
# the blocked triangular solves alone (panel launch, recursion, triangular inverse) and the factorisation against its batch size
(timeout -k 10 200 python3 tools/trsm_bench.py; timeout -k 10 200 python3 tools/potrf_batch.py) > $OUT/${TAG}_trsm_potrf_bench.txt 2>&1 || exit 18
# randomised sweeps of the device fit against the oracle (shapes, iteration counts, kernels, precisions, loss traces)
timeout -k 10 400 python3 tests/dev/fuzz_fit.py 31 50 > $OUT/${TAG}_fuzz_seed31.log 2>&1 || exit 20
timeout -k 10 400 python3 tests/dev/fuzz_fit.py 32 50 > $OUT/${TAG}_fuzz_seed32.log 2>&1 || exit 21
timeout -k 10 400 python3 tests/dev/fuzz_fit.py 47 50 > $OUT/${TAG}_fuzz_seed47.log 2>&1 || exit 22
for s in 101 202 303 404; do timeout -k 10 400 python3 tests/dev/fuzz_fit.py $s 60 > $OUT/${TAG}_fuzz_seed$s.log 2>&1 || exit 29; done
# round 4: the fused 256-wide diagonal-block Cholesky by wave, the fp64 VALU rates the small kernel is priced against, the
# small kernel's reduced export, and the config-5 API path stage by stage
timeout -k 10 120 python3 tools/potrf256_phases.py > $OUT/${TAG}_potrf256_phases.txt 2>&1 || exit 23
timeout -k 10 120 python3 tools/valu64_bench.py > $OUT/${TAG}_valu64_bench.txt 2>&1 || exit 24
timeout -k 10 200 python3 tools/small_phases.py 86 8 8192 reduced > $OUT/${TAG}_small_phases_reduced.txt 2>&1 || exit 25
timeout -k 10 300 python3 tools/api_path_c5_profile.py > $OUT/${TAG}_api_path_c5_profile.txt 2>&1 || exit 26
timeout -k 10 300 python3 tools/scale_check.py > $OUT/${TAG}_scale_check.txt 2>&1 || exit 27
(timeout -k 10 100 python3 tools/worker_bench.py; python3 tools/flow_model.py) > $OUT/${TAG}_worker_bench.txt 2>&1 || exit 28
fi
head -8 $OUT/${TAG}_kernel_stats_final.txt; cat $OUT/${TAG}_pmc_traffic.json | head -12; cut -c1-300 $OUT/${TAG}_bench_final.json
