mkdir -p gpurun_out/r02_probe
{
python3 -c "import cv2; print('cv2', cv2.__version__)" 2>&1 | tail -1
find / -name 'libopencv*' 2>/dev/null | head
find / -name 'cv2*' -maxdepth 6 2>/dev/null | head
ldconfig -p | grep -i opencv | head
pkg-config --modversion opencv4 2>&1 | head -1
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > gpurun_out/r02_probe/opencv_probe.txt 2>&1
python3 bench.py > gpurun_out/r02_probe/bench_base.json 2>gpurun_out/r02_probe/bench_base.err
cat gpurun_out/r02_probe/opencv_probe.txt
cat gpurun_out/r02_probe/bench_base.json
