# what the KFD topology of the box says about the GPU's caches (the library sizes its batch chunks from the level-3 entry)
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  echo "== $n"; grep -E "simd_count|location_id|domain|gfx_target|unique_id" $n/properties 2>&1 | head -8
  ls $n/caches 2>/dev/null | wc -l
  for c in $n/caches/*; do l=$(grep -E "^level" $c/properties 2>/dev/null | awk '{print $2}'); s=$(grep -E "^size" $c/properties 2>/dev/null | awk '{print $2}'); echo "$l $s"; done | sort | uniq -c | head
done
python -c "
import torch
p=torch.cuda.get_device_properties(0); print(p)
"
