"""Where the one-pass FASTA packer's time goes: the kernel cut short at its phases (MM_FASTA_DEBUG, wrong results)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MM_ENV_DYNAMIC", "1")
import simd_minimizers_amd as sm
from simd_minimizers_amd import workloads
dev = torch.device("cuda:0")
ws = sm.Workspace(0, torch.cuda.current_stream(dev).cuda_stream)
os.environ["MM_FASTA_ONEPASS"] = "1"
for dbg, what in ((0, "whole call"), (1, "no look-backs"), (3, "no look-backs, no B"), (9, "no look-backs, stop before A2's scans"),
                  (4, "A1 only"), (2, "look-backs, no B"), (16, "look-back without polling: one round trip")):
    os.environ["MM_FASTA_DEBUG"] = str(dbg)
    try:
        r = workloads.measure("FASTA", ws, dev, warm=3, reps=7)
        print(f"MM_FASTA_DEBUG={dbg} ({what}): {r['ms']:.3f} ms", flush=True)
    except Exception as e:
        print(f"MM_FASTA_DEBUG={dbg} ({what}): {e}", flush=True)
