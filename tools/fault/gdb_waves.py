"""rocgdb script over a GPU core dump (tools/fault/probe.sh): one line per wave (pc, exec, trap status, the instruction at pc), then
for every wave with a memory violation recorded in its trap status - or, when that register cannot be read, for every wave - the scalar
registers and the code around pc."""
import gdb

gdb.execute("set pagination off")
gdb.execute("set width 0")
for cmd in ("info agents", "info queues", "info dispatches"):
    try:
        print("====", cmd)
        gdb.execute(cmd)
    except gdb.error as e:
        print("(failed)", e)


def reg(name):
    try:
        return int(gdb.parse_and_eval("$" + name))
    except Exception:
        return None


inf = gdb.selected_inferior()
threads = sorted(inf.threads(), key=lambda t: t.num)
print("==== waves", len(threads))
suspects = []
for t in threads:
    try:
        t.switch()
    except Exception as e:
        print("thread", t.num, "cannot be selected", e)
        continue
    pc, ex, ts, st = reg("pc"), reg("exec"), reg("trapsts"), reg("status")
    try:
        ins = gdb.execute("x/i $pc", to_string=True).strip()
    except gdb.error as e:
        ins = "(no code at pc: %s)" % e
    viol = ts is not None and (ts & (1 << 8)) != 0
    print("wave %3d %-34s pc=%s exec=%s trapsts=%s status=%s %s%s" % (
        t.num, t.name or "", hex(pc) if pc is not None else None, hex(ex) if ex is not None else None,
        hex(ts) if ts is not None else None, hex(st) if st is not None else None, ins, "   <== MEM_VIOL" if viol else ""))
    if viol or ts is None:
        suspects.append(t)
print("==== suspects", [t.num for t in suspects])
for t in suspects[:12]:
    t.switch()
    print("==== wave", t.num, t.name)
    for cmd in ("bt 4", "info registers pc exec vcc m0 flat_scratch", "x/48i $pc-128"):
        try:
            gdb.execute(cmd)
        except gdb.error as e:
            print("(failed)", cmd, e)
    line = []
    for k in range(0, 104):
        v = reg("s%d" % k)
        line.append("s%d=%s" % (k, hex(v) if v is not None else "?"))
    print(" ".join(line))
    for k in list(range(0, 48)) + [252, 253, 254, 255]:
        try:
            print("v%d" % k, gdb.execute("p/x $v%d" % k, to_string=True).strip())
        except gdb.error as e:
            print("v%d" % k, "(failed)", e)
