#!/bin/bash
# A/B builds of the GEMM kernel only: swift_amd/csrc/variants/libswiftk_<name>.so = gemm.hip compiled with the given
# defines + the product's other objects.  usage: tools/build_variants.sh name "-DFOO=1 -DBAR=0" [name2 "..."] ...
set -e
cd "$(dirname "$0")/../swift_amd/csrc"
make -s -j8
mkdir -p variants
while [ $# -gt 0 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize $defs -c gemm.hip -o variants/gemm_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libswiftk_$name.so variants/gemm_$name.o gemm_tn.o gemm_rownorm.o attention.o attention_pipe.o qkv_attn.o attention_bwd.o elementwise.o train_kernels.o jvp_kernels.o forward.o
  echo "built variants/libswiftk_$name.so ($defs)"
done
