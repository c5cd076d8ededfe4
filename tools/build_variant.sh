#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags...]  -> pylrbms_amd/_variants/NAME.so (travels with gpurun, git-ignored)
# Only fused.hip is recompiled with the extra flags; the other objects come from the regular build.
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from pylrbms_amd._build import build_native; build_native()"
mkdir -p $ROOT/pylrbms_amd/_variants /tmp/var_$NAME
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DLRBMS_EXPERIMENT_BUILD "$@" -I$ROOT/pylrbms_amd/csrc -c ${SRC:-$ROOT/pylrbms_amd/csrc/fused.hip} -o /tmp/var_$NAME/fused.o
OBJS=""
for f in capi assemble apply gemm online enrich fom lrbms3d; do OBJS="$OBJS $ROOT/pylrbms_amd/csrc/_obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/pylrbms_amd/_variants/$NAME.so $OBJS /tmp/var_$NAME/fused.o -L/opt/rocm/lib -lrocsolver -lrocblas
echo built $ROOT/pylrbms_amd/_variants/$NAME.so
