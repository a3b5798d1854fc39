# partition times of the demo frame (and the 100 000-entity frame) under dealing-order variants; PT_TAIL_TILES: tiles dealt range by range
set -e
for tail in default 0 100000; do
  if [ $tail = default ]; then unset PT_TAIL_TILES; else export PT_TAIL_TILES=$tail; fi
  echo "== tail $tail" 
  timeout -k 10 120 python profiles/partition_time.py demo-1080p-64spp-d8 ${RANGES:-0 4 10} 2>/dev/null
done
unset PT_TAIL_TILES
echo "== dense"
timeout -k 10 300 python profiles/partition_time.py dense100k-4k-256spp-d8 0 2>/dev/null
