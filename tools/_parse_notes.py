import re, subprocess, sys
txt = sys.stdin.read()
# the notes are YAML-ish; split per kernel on "- .agpr_count" / ".args"
blocks = re.split(r"\n\s+- \.agpr_count:", txt)
rows = []
for b in blocks[1:]:
    def get(k):
        m = re.search(r"\." + k + r":\s+(\S+)", b)
        return m.group(1) if m else "?"
    name = get("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name)[:70]
    rows.append((name, get("vgpr_count"), get("sgpr_count"), get("group_segment_fixed_size"), get("vgpr_spill_count"), get("private_segment_fixed_size")))
for r in rows:
    print(f"{sys.argv[1]:22s} {r[0]:70s} vgpr {r[1]:>4s} sgpr {r[2]:>4s} lds {r[3]:>6s} spill {r[4]:>3s} scratch {r[5]:>4s}")
