#!/bin/bash
# per-role register demand of a role-split kernel: compiles the unit with `role` pinned to each value in turn (the other branches
# fold away) and prints vgpr_count / vgpr_spill_count.  usage: tools/exp/role_regs.sh train_bwd_h3t.hip "const int role = wave / 3, rw = wave - role \* 3;" "const int role = %d, rw = wave % 3;" 4
src=/root/repo/blind_image_denoising_amd/csrc/$1
for r in $(seq 0 $(($4 - 1))); do
  rep=$(printf "$3" $r)
  sed "s/$2/$rep/" $src > /tmp/only_role.hip
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I/root/repo/blind_image_denoising_amd/csrc -Wno-unused-function -Wno-pass-failed -c /tmp/only_role.hip -o /tmp/only_role.o -save-temps=obj 2>&1 | grep -E "error" -A3 | head
  echo "role $r: $(grep -E '^\s+\.(vgpr_count|vgpr_spill_count):' /tmp/only_role-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | head -1)"
done
