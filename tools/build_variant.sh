#!/bin/bash
# tools/build_variant.sh NAME "-DDAL3_ENC_T=2 ..."  -> variants/NAME.so (for tools/ab_kernels.py)
# mirrors 3dal_pytorch_amd/csrc/Makefile (the 16-bit kernels are compiled twice, see dal3_pointmlp_lp.hip)
set -e
cd "$(dirname "$0")/.."
mkdir -p variants/obj_$1
CC="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
NONAN=-fno-honor-nans          # the shared-MLP kernels only, as in the Makefile
for f in dal3_api dal3_misc dal3_prep dal3_crops dal3_train dal3_train_fc; do
  $CC $2 -c 3dal_pytorch_amd/csrc/$f.hip -o variants/obj_$1/$f.o &
done
$CC $2 -mllvm -pragma-unroll-threshold=1000000 -c 3dal_pytorch_amd/csrc/dal3_train_x3.hip -o variants/obj_$1/dal3_train_x3.o &
CC="$CC $NONAN"
$CC $2 -c 3dal_pytorch_amd/csrc/dal3_pointmlp.hip -o variants/obj_$1/dal3_pointmlp.o &
$CC $2 -c 3dal_pytorch_amd/csrc/dal3_latency.hip -o variants/obj_$1/dal3_latency.o &
X3="-mllvm -pragma-unroll-threshold=1000000"
$CC $2 $X3 -DX3_PART=1 ${X3_ENC_FORM--mllvm -amdgpu-mfma-vgpr-form} -c 3dal_pytorch_amd/csrc/dal3_pointmlp_x3.hip -o variants/obj_$1/x3_enc.o &
$CC $2 $X3 -DX3_PART=2 -c 3dal_pytorch_amd/csrc/dal3_pointmlp_x3.hip -o variants/obj_$1/x3_dec.o &
$CC $2 -DLP_PART=1 -mllvm -amdgpu-mfma-vgpr-form -c 3dal_pytorch_amd/csrc/dal3_pointmlp_lp.hip -o variants/obj_$1/lp_enc.o &
$CC $2 -DLP_PART=2 -c 3dal_pytorch_amd/csrc/dal3_pointmlp_lp.hip -o variants/obj_$1/lp_dec.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 variants/obj_$1/*.o -o variants/$1.so
rm -rf variants/obj_$1
echo built variants/$1.so
