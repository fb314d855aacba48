"""The kernels of ONE network forward of the C3 iteration, in launch order with their durations, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline
    python3 tools/kernel_sequence.py /tmp/tr
(a forward = the launches from one graph build -- its first kernel -- to the next)."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
marks = [i for i, (n, _) in enumerate(names) if "egnn_graph_mask_kernel" in n]
spans = [(a, b) for a, b in zip(marks, marks[1:]) if b - a > 10]
# the shortest one: a forward of the default (split-f16) mode, not of the exact-f32 mode bench.py times beside it
a, b = min(spans, key=lambda ab: sum(us for _, us in names[ab[0]:ab[1]]))
total = 0.0
for n, us in names[a:b]:
    short = re.sub(r"\(anonymous namespace\)::", "", n)
    short = re.sub(r"^void ", "", short)
    # our kernels: the name up to the argument list; library kernels carry what they do inside their template arguments
    short = short[:200] if short.startswith(("at::", "rocprim::")) else re.split(r"\(", short)[0][:95]
    print(f"{us:9.1f} us  {short}")
    total += us
print(f"{total:9.1f} us in {b - a} kernels")
