#!/bin/bash
# GPU box: the stream of 4096-batches in the reference order under launch widths / hand-over settings (developer scan)
for c in 2 3; do
  for v in "DFTPAV_REF_SLOTS=512" "DFTPAV_REF_SLOTS=768" "DFTPAV_REF_SLOTS=1024" "DFTPAV_STREAM_HAND_OVER=-1 DFTPAV_REF_QUAD_HANDOVER=128" "DFTPAV_STREAM_HAND_OVER=-1 DFTPAV_REF_QUAD_HANDOVER=384" "DFTPAV_STREAM_HAND_OVER=-1 DFTPAV_REF_QUAD_HANDOVER=768"; do
    echo "$v"; env $v CFG=$c timeout 300 python scripts/ref_stream_time.py ${DEPTHS:-4}
  done
done 2>&1 | grep -v "Warning\|amdgpu.ids"
