import re
def edit(path, pairs):
    """paragraph- and list-item-aware replace: a unit is a table row, or a list item, or a blank-line separated paragraph;
    inside a unit whitespace runs (line wraps) are flattened before matching, tools/wrap_md.py rewraps afterwards"""
    lines=open(path).read().split("\n")
    units=[]; cur=[]
    def flush():
        if cur: units.append(list(cur)); cur.clear()
    for ln in lines:
        if ln.strip()=="" :
            flush(); units.append([ln]); continue
        if ln.startswith("|") or ln.startswith("#") or ln.startswith("```"):
            flush(); units.append([ln]); continue
        if ln.startswith("- "):
            flush(); cur.append(ln); continue
        cur.append(ln)
    flush()
    for old,new in pairs:
        hit=False
        for i,u in enumerate(units):
            if len(u)==1:
                if old in u[0]:
                    units[i]=[u[0].replace(old,new)]; hit=True; break
                continue
            flat=u[0]+"".join(" "+x.strip() for x in u[1:])
            if old in flat:
                units[i]=[flat.replace(old,new)]; hit=True; break
        assert hit, old[:90]
    open(path,'w').write("\n".join("\n".join(u) for u in units))
