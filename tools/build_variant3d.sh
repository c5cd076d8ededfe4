#!/bin/bash
# usage: tools/build_variant3d.sh NAME [extra hipcc flags...]  -> pylrbms_amd/_variants/NAME.so: lrbms3d.hip recompiled with the extra flags,
# the other objects from the regular build (the 3D counterpart of tools/build_variant.sh)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from pylrbms_amd._build import build_native; build_native()"
mkdir -p $ROOT/pylrbms_amd/_variants /tmp/var_$NAME
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DLRBMS_EXPERIMENT_BUILD "$@" -I$ROOT/pylrbms_amd/csrc -c $ROOT/pylrbms_amd/csrc/lrbms3d.hip -o /tmp/var_$NAME/lrbms3d.o
OBJS=""
for f in capi assemble apply gemm online enrich fom fused; do OBJS="$OBJS $ROOT/pylrbms_amd/csrc/_obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/pylrbms_amd/_variants/$NAME.so $OBJS /tmp/var_$NAME/lrbms3d.o -L/opt/rocm/lib -lrocsolver -lrocblas
echo built $ROOT/pylrbms_amd/_variants/$NAME.so
