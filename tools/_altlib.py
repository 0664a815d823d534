"""tools only: OSK_ALT_LIB=<path to an alternative libosk_hip.so> (tools/make_attn_variants.sh, tools/make_ablated_libs.sh) makes
`open_sora_amd._C` load that library instead of the shipped one.  Import BEFORE anything imports open_sora_amd._C."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def install():
    alt = os.environ.get("OSK_ALT_LIB")
    if not alt:
        return None
    spec = importlib.util.spec_from_file_location("open_sora_amd.build", os.path.join(ROOT, "open_sora_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.LIB_PATH = alt if os.path.isabs(alt) else os.path.join(ROOT, alt)
    b.build_lib = lambda *a, **k: b.LIB_PATH
    sys.modules["open_sora_amd.build"] = b
    return b.LIB_PATH
