import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
marks = [i for i, r in enumerate(rows) if "k_preprocess" in r[2]]
win = rows[marks[-2]:marks[-1]]
c = collections.Counter()
for s, e, n in win:
    if "at::native" in n:
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda\w*|direct_copy\w*|lambda[^>]*)", n)
        c[n[:60] + " ... " + " | ".join(re.findall(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n)[1:4])] += 1
for k, v in c.most_common(30): print(v, k)
