#!/bin/bash
# Profile passes behind profiles/rNN_{kernel_stats,pmc_traffic,pmc_sq}_*: one rocprofv3 run per counter group (never --pmc together with
# a trace domain other than --kernel-trace), each of `bench.py --steps 2 --warmup 1` with the legs that do not belong to the
# configuration switched off.  Usage (on the GPU box, from the repository root):  bash tools/profile_round.sh <out_dir> [cfg ...]
#   cfg: c2 (Bernoulli 1e7 x 512) | m1024 (Bernoulli 1e7 x 1024) | c3r (NegBin r = 15, one rank's share 1.25e6 x 1024) | c4 (categorical K = 10, 1e6 x 256)
# Post-processing (anywhere): tools/pmc_traffic_json.py <fetch> <write> <stats csv> <out.json> N M lik L ; tools/pmc_sq_json.py <dir> <out.json>
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
export TMPDIR=/tmp
O=$ROOT/$1; shift
mkdir -p "$O"
COMMON="--steps 2 --warmup 1 --no-cpu --no-parity --no-m1024 --no-c5 --no-f32 --no-elbo --no-extra"
for cfg in "$@"; do
  case $cfg in
    c2) ARGS="$COMMON";;
    m1024) ARGS="$COMMON --inducing 1024 --no-gibbs";;
    c3r) ARGS="$COMMON --lik negbin --points 1250000 --inducing 1024";;
    c4) ARGS="$COMMON --lik categorical --points 1000000 --inducing 256";;
    *) echo "unknown cfg $cfg"; continue;;
  esac
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${cfg}_stats" -- python3 "$ROOT/bench.py" $ARGS > "$O/${cfg}_stats.log" 2>&1; echo "rc $cfg stats $?"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/${cfg}_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$O/${cfg}_fetch.log" 2>&1; echo "rc $cfg fetch $?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/${cfg}_write" -- python3 "$ROOT/bench.py" $ARGS > "$O/${cfg}_write.log" 2>&1; echo "rc $cfg write $?"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/${cfg}_sq" -- python3 "$ROOT/bench.py" $ARGS > "$O/${cfg}_sq.log" 2>&1; echo "rc $cfg sq $?"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/${cfg}_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$O/${cfg}_sq2.log" 2>&1; echo "rc $cfg sq2 $?"
  cd "$ROOT"
done
du -sh "$O"
