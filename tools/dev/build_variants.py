"""dev: build named -DLBMPM_DEV variants of the library side by side:  python tools/dev/build_variants.py name=-DA,-DB=2 name2=...
-> tools/dev/_build/lib_<name>.so (an empty flag list builds the plain development flavour)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from openlbmpm_amd import build
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    extra = [f for f in flags.split(",") if f]
    out = os.path.join(ROOT, "tools", "dev", "_build", "lib_%s.so" % name)
    build.build_dev(out, verbose=False, extra=extra)
    print(out, flush=True)
