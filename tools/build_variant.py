"""Build a variant of libnaf_hip.so for an interleaved A/B on one lease: python tools/build_variant.py OUT.so [-DFLAG ...].
Objects go to a scratch directory, the product library is untouched; run with NAF_HIP_LIB=OUT.so."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from naf_amd import build as B

out = os.path.abspath(sys.argv[1])
B.OBJDIR = tempfile.mkdtemp(prefix="naf_variant_")
B.LIB = out
B.build_library(force=True, verbose=True, extra_flags=sys.argv[2:])
