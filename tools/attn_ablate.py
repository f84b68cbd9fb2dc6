"""Compile attention.hip with ATT_ABL ablation masks into tools/tmp_libs/ (run here, on CPU), then time each build
on the GPU:   python tools/attn_ablate.py build ; gpurun -- python tools/attn_ablate.py run
Mask bits: 1 no v_exp, 2 no K/V loads in the loop, 4 no barrier, 8 no softmax VALU at all."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "imagharmony_amd", "csrc"); OBJ = os.path.join(CSRC, "_obj"); TMP = os.path.join(ROOT, "tools", "tmp_libs")
MASKS = [int(x) for x in os.environ.get("ATT_MASKS", "0,1,8,2,6,14").split(",")]
SRC = os.environ.get("ATT_SRC", os.path.join(CSRC, "attention.hip"))
if sys.argv[1] == "build":
    os.makedirs(TMP, exist_ok=True)
    others = [os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".o") and not f.startswith("attention")]
    for m in MASKS:
        o = os.path.join(TMP, f"attn_abl{m}.o")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DATT_ABL={m}", "-I", CSRC, "-c", SRC, "-o", o], check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(TMP, f"libimh_abl{m}.so"), o] + others, check=True)
        os.remove(o)
    print("built", MASKS)
elif sys.argv[1] == "run":
    for m in MASKS:
        env = dict(os.environ, IMH_LIB_PATH=os.path.join(TMP, f"libimh_abl{m}.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_bench.py")], env=env, capture_output=True, text=True).stdout
        print(f"== ATT_ABL={m}"); print("\n".join(l for l in out.splitlines() if l.startswith("self")), flush=True)
