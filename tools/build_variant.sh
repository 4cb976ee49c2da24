#!/bin/bash
# tools/build_variant.sh NAME "-DDAL3_ENC_T=2 ..."  -> variants/NAME.so (for tools/ab_kernels.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$1
for f in dal3_api dal3_pointmlp dal3_pointmlp_lp dal3_misc dal3_prep; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-honor-nans $2 \
      -c 3dal_pytorch_amd/csrc/$f.hip -o variants/obj_$1/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 variants/obj_$1/*.o -o variants/$1.so
rm -rf variants/obj_$1
echo built variants/$1.so
