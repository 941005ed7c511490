#!/bin/bash
# where the fused FFN kernel's time goes: builds without the GELU math / without the K loop (tuning only)
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT/tristage-rag_amd/csrc
SRCS="ts_index.hip ts_scan.hip ts_scan_f32s.hip ts_fused.hip ts_select.hip ts_maxsim.hip ts_maxsim16.hip ts_bm25.hip ts_fwd.hip ts_ffn.hip"
for v in base nogelu noloop; do
  case $v in base) F="";; nogelu) F="-DFF_NO_GELU";; noloop) F="-DFF_NO_LOOP";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $F -shared -o ../variants_$v.so $SRCS &
done
wait
cd $ROOT
for v in base nogelu noloop; do
  echo "== $v"; TRISTAGE_LIB=$ROOT/tristage-rag_amd/variants_$v.so timeout -k 10 120 python tools/ffn_probe.py 2>/dev/null | head -1 | cut -c1-260
done
rm -f tristage-rag_amd/variants_*.so
