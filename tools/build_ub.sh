#!/bin/bash
# Builds the kernel microbenchmarks (tools/ub_*.hip: one kernel header + a main) for gfx950 into tools/_ub/ (git-ignored; the
# binaries travel to the GPU box with the repository snapshot).  Same flags as peppa_pig_face_landmark_amd/build.py.
#   bash tools/build_ub.sh [name ...]        e.g.  bash tools/build_ub.sh ub_sepup     then     gpurun -- 'bash tools/gpu_ub.sh ub_sepup 256 r06_runN'
# -DPF_ABLATE=1 flavours (timing ablations, SepupArgs::dbg): ub_sepup_ablate; -DFRONT2_ABL=n: ub_front2_<n>.
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_ub
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -fma-mix-insts -I peppa_pig_face_landmark_amd/csrc"
names=${*:-"ub_sepup ub_fc2 ub_front2"}
for n in $names; do
  /opt/rocm/bin/hipcc $FLAGS tools/$n.hip -o tools/_ub/$n 2>&1 | grep -v "not a recognized feature" 
  echo "built tools/_ub/$n"
done
if [[ " $names " == *" ub_sepup "* ]]; then /opt/rocm/bin/hipcc $FLAGS -DPF_ABLATE=1 tools/ub_sepup.hip -o tools/_ub/ub_sepup_ablate 2>&1 | grep -v "not a recognized feature"; fi
