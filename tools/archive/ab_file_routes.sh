#!/bin/bash
# NOTE (round 5): MODGPU_HOST_LANES / _RAMP_KB / _SPLIT / _CHUNK_MIN_MB / _FILE_LANES are no longer read by libmodgpu.so (modbench links the
# shipped library); this script documents how profiles/r04_file_routes.txt was taken and does not reproduce it on a round-5 build.
# tools/archive/ab_file_routes.sh -- the file routes (bin/modbench --files) under round 3's staging settings and under this round's defaults,
# interleaved three times (tmpfs writes vary by +-30 % between runs of one command): did the new chunking / ramp / lanes cost the
# file endpoints anything?  Run on the GPU box from the repo root.
O=gpurun_out/ab_files
rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  MODGPU_HOST_LANES=0 MODGPU_HOST_RAMP_KB=0 MODGPU_HOST_SPLIT=16 MODGPU_HOST_CHUNK_MIN_MB=4 modulate_amd/bin/modbench --files /dev/shm --bytes 411000000 --bytes 4294967296 > $O/r3_$rep.txt
  modulate_amd/bin/modbench --files /dev/shm --bytes 411000000 --bytes 4294967296 > $O/r4_$rep.txt
  MODGPU_HOST_FILE_LANES=2 modulate_amd/bin/modbench --files /dev/shm --bytes 411000000 --bytes 4294967296 > $O/l2_$rep.txt
  MODGPU_HOST_FILE_LANES=4 modulate_amd/bin/modbench --files /dev/shm --bytes 411000000 --bytes 4294967296 > $O/l4_$rep.txt
done
for tag in r3 r4; do
  echo "== $tag settings, three runs: route GB/s (file->file, file->pinned, pinned->file) at 392 MiB | 4096 MiB"
  for rep in 1 2 3; do
    grep "modgpu_cycle_" $O/${tag}_$rep.txt | awk '{printf "%s ", $(NF>0? ( ($2=="(file") ? 5 : 5 ) : 0)}' ; echo
  done
done
grep -h "modgpu_cycle_" $O/*.txt | cut -c1-75 | sort | uniq -c | head -0
for f in $O/*.txt; do echo "-- $f"; grep "modgpu_cycle_" $f | cut -c1-62; done
