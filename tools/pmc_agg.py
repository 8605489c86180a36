import csv, glob, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0][-60:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,d in acc.items():
    if "gemm" not in k: continue
    print(k, {c: round(v/cnt[(k,c)]) for c,v in d.items()})
