"""dev: build a variant of the product library with extra -D switches:  python tools/dev/variant.py TAG [-DX ...]  -> tools/dev/_build/lib_TAG.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openlbmpm_amd import build
tag, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "dev", "_build", "lib_%s.so" % tag)
os.makedirs(os.path.dirname(out), exist_ok=True)
build._compile_and_link(out, extra, os.path.join(ROOT, "tools", "dev", "_build", "obj_" + tag), False)
print(out)
