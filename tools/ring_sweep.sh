# B = 100 step against the ring's size and event spacing (IrtEngine.rows_ring_slots / rows_ring_event_every; needs the two read
# from VX_RING_SLOTS / VX_RING_EVERY, as they were while this was measured)
set -e
mkdir -p gpurun_out/r5y
for cfg in "8 4" "8 2" "4 2" "16 4" "16 8" "32 8" "8 1"; do
  set -- $cfg
  for rep in 1 2 3; do
    echo "slots $1 every $2 rep $rep" >> gpurun_out/r5y/mb.log
    VX_RING_SLOTS=$1 VX_RING_EVERY=$2 timeout -k 10 120 python tools/minibatch_probe.py --modes graph --steps 1000 2>&1 | grep graph >> gpurun_out/r5y/mb.log
  done
done
cat gpurun_out/r5y/mb.log
