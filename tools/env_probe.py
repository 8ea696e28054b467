"""Environment probe: what hardware / software the stack is running on.
Run on the GPU box: ``python tools/env_probe.py > gpurun_out/env_probe.md``."""
import os
import platform
import shutil
import subprocess
import sys


def sh(cmd, limit=60):
    try:
        out = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=60).stdout.strip().splitlines()
        return "\n".join(out[:limit])
    except Exception as e:
        return f"(failed: {e})"


def main():
    import torch
    print("# Environment probe\n")
    print(f"- python {sys.version.split()[0]} on {platform.platform()}")
    print(f"- torch {torch.__version__}, hip {torch.version.hip}, cuda.is_available={torch.cuda.is_available()}, "
          f"device_count={torch.cuda.device_count()}")
    print(f"- hipcc: {shutil.which('hipcc')}; HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        free, total = torch.cuda.mem_get_info()
        print(f"- device 0: {p.name}, gcnArch={getattr(p, 'gcnArchName', '?')}, CUs={p.multi_processor_count}, "
              f"HBM total={total / 2**30:.1f} GiB free={free / 2**30:.1f} GiB, warp_size={getattr(p, 'warp_size', '?')}")
    print("\n## rocminfo (agents)\n```")
    print(sh("rocminfo | grep -E 'Marketing Name|Name: +gfx|Compute Unit|Wavefront Size|Max Clock' | head -40"))
    print("```\n## rocm-smi topology\n```")
    print(sh("rocm-smi --showtopo 2>/dev/null | head -40"))
    print("```")


if __name__ == "__main__":
    main()
