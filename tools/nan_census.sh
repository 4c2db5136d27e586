#!/bin/bash
# NaN census of the Stage-I step over launch modes / switches: one arm per process.
# usage: tools/nan_census.sh OUTDIR [STEPS] [REPEATS] [ARMS...]   (default arms: all)
out=${1:-gpurun_out/census}; steps=${2:-150}; reps=${3:-10}; shift 3
arms="${*:-two_stream one_stream hybrid no_fused no_transpose gate_skip det_two snap_two}"
mkdir -p "$out"
{ hostname; rocm-smi --showuniqueid 2>/dev/null | grep -i "GPU\[" | head -1; } > "$out/box.txt" 2>&1; cat "$out/box.txt" >&2
run() {  # tag, env assignments..., -- tool args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  case " $arms " in *" $tag "*) ;; *) return;; esac
  echo "== $tag" >&2
  env "${envs[@]}" timeout -k 10 600 python tools/nan_hunt.py --steps "$steps" --repeats "$reps" --tag "$tag" "$@" \
      > "$out/$tag.jsonl" 2> "$out/$tag.err" || echo "rc=$? ($tag)" >&2
  grep '"summary"' "$out/$tag.jsonl" >&2
}
run two_stream   X=1 -- --mode eager --diag
run one_stream   FMRI_SIDE_STREAM=off -- --mode eager --diag
run hybrid       X=1 -- --mode hybrid --diag
run graph        X=1 -- --mode graph --diag
run no_fused     FMRI_FUSED_APPLY=off -- --mode eager
run no_transpose FMRI_PACK_TRANSPOSE=off -- --mode eager
run gate_skip    X=1 -- --mode eager --gate-skip --diag
run det_two      X=1 -- --mode eager --det --repeats 3 --diag
run snap_two     X=1 -- --mode eager --diag --snap
run snap_hybrid  X=1 -- --mode hybrid --diag --snap
run wl_stage2     X=1 -- --workload stage2
run wl_dual1      X=1 -- --workload dual1
run wl_stage3px   X=1 -- --workload stage3_px128
