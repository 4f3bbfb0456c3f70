"""Developer A/B: build a second copy of the library with extra compiler flags.
    python tools/build_variant.py <name> [-DFLAG ...]   ->  craft_amd/_variants/libcraft_hip_<name>.so
Run with  CRAFT_HIP_LIB=craft_amd/_variants/libcraft_hip_<name>.so  (craft_amd/hip.py) on the same GPU box as the default build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from craft_amd import build as B


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    root = os.path.join(os.path.dirname(B.LIB), "_variants")
    os.makedirs(root, exist_ok=True)
    B.OBJ = os.path.join(root, "obj_" + name)
    B.LIB = os.path.join(root, f"libcraft_hip_{name}.so")
    print(B.build_extension(force=True, extra_flags=flags))


if __name__ == "__main__":
    main()
