"""Rebuilds the variant of the fused step kernel that aborted in round 2 (gpurun_out/ms_tests.log: the step body as a `noinline`
device function called from the kernel) and prints its resource usage -- compile only, nothing is launched.
    python scripts/diag_noinline_variant.py
Finding: 128 VGPRs and the same 896 B of static LDS as the kernel proper, but ~1980 bytes of SCRATCH per lane: the
by-value argument block is copied to every lane's private stack so that the callee can take its address; the kernel proper uses none."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REPO, "rlao_amd", "csrc", "step_kernel.hip")).read()
old = "template <int KS, bool PE>\n__global__ void __launch_bounds__(1024) k_env_step_sh6(const StepArgs a, const StepLds L) {"
assert src.count(old) == 1
src = src.replace(old, "template <int KS, bool PE>\n__device__ __attribute__((noinline)) void step_body(const StepArgs& a, const StepLds& L) {")
end = src.index("    AO_WSTAMP(6);\n}\n") + len("    AO_WSTAMP(6);\n}\n")
src = src[:end] + ("\ntemplate <int KS, bool PE>\n__global__ void __launch_bounds__(1024) k_env_step_sh6(const StepArgs a, const StepLds L) {\n"
                   "    step_body<KS, PE>(a, L);\n}\n") + src[end:]
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "step_noinline.hip")
    open(f, "w").write(src)
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{REPO}/include", f"-I{REPO}/rlao_amd/csrc",
                          "-c", f, "-o", os.path.join(d, "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
show = False
for line in out.splitlines():
    if "Function Name" in line:
        show = "k_env_step_sh6ILi6ELb0" in line or "step_bodyILi6ELb0" in line
    if show and re.search(r"Function Name|VGPRs:|ScratchSize|Dynamic Stack|LDS Size", line):
        print(line.split("remark:")[-1].strip())
