#!/bin/bash
# ~8 s of the saturating fp64 MFMA loop with rocm-smi power / clock samples taken back to back DURING it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/power; mkdir -p $O
( timeout 60 tools/bin/mfma_clock_bench 8 > $O/mfma_long.txt 2>&1 ) &
LONG=$!
sleep 1.0
( while kill -0 $LONG 2>/dev/null; do /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; done ) > $O/smi_during.jsonl
wait $LONG
( for i in 1 2 3; do sleep 1; /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; done ) > $O/smi_idle.jsonl
python - <<PY > $O/r02_mfma_power_samples.txt 2>&1
import json
def rows(path):
    out=[]
    for line in open(path):
        try: d=json.loads(line)
        except Exception: continue
        for card,v in d.items():
            if isinstance(v,dict):
                out.append({k.strip(':'):v[k] for k in v if any(s in k.lower() for s in ("power","sclk"))})
    return out
print("rocm-smi --showpower --showclocks samples taken back to back WHILE tools/bin/mfma_clock_bench ran its saturating")
print("configuration (v_mfma_f64_16x16x4_f64, 8 accumulators, 2 waves / SIMD) for ~8 s:")
for r in rows("$O/smi_during.jsonl"): print(" ", r)
print("idle, 1-3 s after the loop:")
for r in rows("$O/smi_idle.jsonl"): print(" ", r)
print("the loop's own report (last lines; tick rate = shader clock seen by s_memtime / wall):")
for t in open("$O/mfma_long.txt").read().strip().splitlines()[-3:]: print(" ", t)
PY
cat $O/r02_mfma_power_samples.txt
