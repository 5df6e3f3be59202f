"""Where things live (shared by the build script and the ctypes loader)."""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
# CGPT_LIB_PATH: load another build of the library (A/B runs of two builds in one GPU session); the default is the in-tree build
LIB_PATH = os.environ.get("CGPT_LIB_PATH") or os.path.join(LIB_DIR, "libcpugpupt.so")
