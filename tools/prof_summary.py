"""compact view of a tools/trace_window.py per-iteration csv: short kernel names, launches, busy us; torch kernels grouped"""
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
tot_busy = tot_launch = 0.0
out, torch_busy, torch_launch = [], 0.0, 0.0
for r in rows[1:]:
    if len(r) < 4:
        print(' '.join(r)); continue
    try:
        n, l, b = r[0], float(r[1]), float(r[2])
    except ValueError:
        print(' '.join(r)); continue
    short = re.sub(r'\(anonymous namespace\)::', '', n)
    short = re.sub(r'^void ', '', short)
    is_torch = short.startswith('at::') or short.startswith('__amd') or 'hipcub' in short or 'rocprim' in short or 'Cijk' in short or 'miopen' in short.lower()
    short = short.split('(')[0][:70] if not is_torch else re.sub(r'at::native::', '', short)[:70]
    tot_busy += b; tot_launch += l
    if is_torch:
        torch_busy += b; torch_launch += l
    out.append((b, l, short, is_torch))
out.sort(reverse=True)
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for b, l, s, t in out[:lim]:
    print(f'{b:9.1f} us {l:6.1f} x  {"[torch] " if t else ""}{s}')
print(f'total busy {tot_busy:.0f} us/iter, {tot_launch:.0f} launches; torch/library kernels {torch_busy:.0f} us, {torch_launch:.0f} launches')
