#!/bin/bash
# round 6, batch h: cross-stream dependency probe (tools/xstream_probe.hip), standalone (/opt/rocm runtime) and under torch's HIP runtime
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r6h; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== standalone" > $O/probe.txt
timeout 120 $R/tools/bin/xstream_probe >> $O/probe.txt 2>&1
echo "== torch runtime" >> $O/probe.txt
timeout 200 python -c "
import torch, ctypes
torch.zeros(1, device='cuda')
l = ctypes.CDLL('$R/tools/bin/libxstream_probe.so')
l.xstream_probe_run()
" >> $O/probe.txt 2>&1
cat $O/probe.txt
