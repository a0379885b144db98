echo "== base (round 4's skinning phase)"; SMPLPP_HIP_LIB=$PWD/ab/base.so timeout -k 10 300 python3 tools/fk_part_ordered.py 2>&1 | grep -v amdgpu.ids
echo "== new"; timeout -k 10 300 python3 tools/fk_part_ordered.py 2>&1 | grep -v amdgpu.ids
