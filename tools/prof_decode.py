import sys,re,ast
for line in sys.stdin:
    if "HIP tie census" in line:
        d=ast.literal_eval(line.split("census",1)[1].strip())
        v=list(d.values())
        print("pts(us):",[x/100 for x in v[:7]], "worst: total %.1f us chunks %.1f us local %.1f us full_sum %.1f us"%((v[7]>>44)/100, ((v[7]>>24)&0xfffff)/100, ((v[7]>>12)&4095)*0.16, (v[7]&4095)*0.16))
