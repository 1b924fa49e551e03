import os, sys, subprocess
for nb in (1024, 2048, 4096, 8192, 16384, 65536):
    out = subprocess.run([sys.executable, "-c", """
import torch, bench
dev=torch.device('cuda:0'); torch.cuda.set_device(0)
print(%d, bench.measured_peaks(dev))
""" % nb], env=dict(os.environ, DN_UBENCH_COPY_BLOCKS=str(nb)), capture_output=True, text=True)
    print(out.stdout.strip(), out.stderr.replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory", "").strip()[-300:])
