"""tools/kernel_sequence.py TRACE_DIR N_LAST -- the last N kernel launches (short names, in order) from a rocprofv3 kernel trace."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
for r in rows[-n:]:
    name = r["Kernel_Name"]
    m = re.search(r"(whvi::\w+|Cijk_\w{0,20}|multi_tensor_apply_kernel|reduce_kernel|FillFunctor|MulFunctor|CUDAFunctor_add|CUDAFunctorOnSelf_add|"
                  r"CUDAFunctorOnOther_add|direct_copy|copyBuffer|clamp|threshold|distribution\w*|sigmoid|neg_kernel|log_kernel|reciprocal|\w+Functor)", name)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{dur:7.2f} us  {m.group(1) if m else name[:60]}")
