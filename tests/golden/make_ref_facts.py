"""Regenerates tests/golden/ref_facts.json:  python tests/golden/make_ref_facts.py  (needs /root/reference)

Facts read off the reference's TEXT -- the one kind of reference-held evidence this image allows (the reference cannot be built: GSL is absent; it
ships no vectors).  Like make_h5_listing.py for the HDF5 layout, this extracts values and orders, no source text:
  constants        the ten physical constants of Src/mclib.c:4-5 (name -> literal as written)
  struct_photon    member names and C types of struct photon in declaration order, thermal-only build (Src/mcrat.h:142-171)
  struct_photon_list   the same for struct photonList (Src/mcrat.h:173-180)
  hot_table_grid   LOG_PH_E_MIN/MAX, N_PH_E, LOG_T_MIN/MAX, N_T (Src/hot_x_section.h:2-10)
  checkpoint_fwrite_order   saveCheckpoint's fwrite calls in the branch that writes photons: (variable, sizeof type) (Src/mcrat_io.c:871-903)
  mcpar_read_order readMcPar's destinations in the order it reads them (Src/mcrat_io.c:1136-1237)
tests/test_ref_facts.py holds the engine's constants, its struct photon layout, its checkpoint writer and its mc.par parser against them."""
import json
import os
import re

REF = "/root/reference/Src"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def struct_members(header, name):
    m = re.search(r"struct\s+%s\s*\{(.*?)\}\s*;" % name, header, flags=re.S)
    body, first = m.group(1), header[:m.start()].count("\n") + 1
    out, skip = [], 0
    for line in body.split("\n"):
        s = line.strip()
        if re.match(r"#if", s):
            skip += 1                       # (NONTHERMAL_E_DIST members: not in the thermal-only build the engine mirrors)
            continue
        if re.match(r"#endif", s):
            skip -= 1
            continue
        if skip or not s:
            continue
        d = re.match(r"((?:struct\s+)?\w+)\s*(\*?)\s*(\w+)\s*(\[[^\]]*\])?\s*;", s)
        if d:
            out.append({"type": d.group(1) + (" *" if d.group(2) else ""), "name": d.group(3)})
    return out, first


def main():
    facts = {"generated_by": "tests/golden/make_ref_facts.py", "reference": "lazzati-astro/MCRaT Src/ (snapshot under /root/reference)"}
    mclib = open(os.path.join(REF, "mclib.c")).read()
    consts = {}
    for line_no, line in enumerate(mclib.split("\n")[:12], 1):
        if line.strip().startswith("const double"):
            for name, val in re.findall(r"(\w+)\s*=\s*([0-9.eE+-]+)", strip_comments(line)):
                consts[name] = {"literal": val, "line": line_no}
    facts["constants"] = consts
    header = strip_comments(open(os.path.join(REF, "mcrat.h")).read())
    raw_header = open(os.path.join(REF, "mcrat.h")).read()
    ph, _ = struct_members(header, "photon")
    facts["struct_photon"] = {"members": ph, "line": raw_header[:raw_header.index("struct photon\n")].count("\n") + 1}
    pl, _ = struct_members(header, "photonList")
    facts["struct_photon_list"] = {"members": pl, "line": raw_header[:raw_header.index("struct photonList")].count("\n") + 1}
    hot = strip_comments(open(os.path.join(REF, "hot_x_section.h")).read())
    facts["hot_table_grid"] = {k: float(v) if "." in v else int(v) for k, v in re.findall(r"#define\s+(LOG_PH_E_MIN|LOG_PH_E_MAX|N_PH_E|LOG_T_MIN|LOG_T_MAX|N_T)\s+(-?[0-9.]+)", hot)}
    io_raw = open(os.path.join(REF, "mcrat_io.c")).read()
    io = strip_comments(io_raw)
    start = io.index("int saveCheckpoint(")
    body = io[start:io.index("\n}", io.index("return success", start))]
    # the first branch (a frame that is neither the last nor the injection frame) writes the full header and the photons
    branch = body[:body.index("else if")]
    facts["checkpoint_fwrite_order"] = [{"variable": v.strip().lstrip("&("), "sizeof": t.strip()} for v, t in re.findall(r"fwrite\s*\(\s*([^,]+),\s*sizeof\s*\(([^)]*)\)", branch)]
    facts["checkpoint_fwrite_line"] = io_raw[:io_raw.index("int saveCheckpoint(")].count("\n") + 1
    start = io.index("void readMcPar(")
    body = io[start:io.index("void dirFileMerge(")]
    order = []
    for m in re.finditer(r"fscanf\s*\(\s*fptr\s*,\s*\"(%\w+)\"\s*,\s*&?\(?\s*&?\(?\(?([\w>\-\*]+)\)?\s*(\[\d\])?|\*\s*(\w+)\s*=\s*getc\s*\(\s*fptr\s*\)|\(\*(\w+)\)\[i\]\s*=\s*strto(l|f)", body):
        if m.group(1):
            dest = m.group(2).replace("hydro_data->", "").lstrip("*") + (m.group(3) or "")
            if dest == "theta_deg":                 # a temporary: the line after names the destination
                nxt = re.search(r"\*(\w+)\s*=\s*theta_deg", body[m.end():])
                dest = nxt.group(1)
            order.append({"dest": dest, "format": m.group(1)})
        elif m.group(4):
            order.append({"dest": m.group(4), "format": "getc"})
        else:
            order.append({"dest": m.group(5) + "[i]", "format": "strto" + m.group(6)})
    facts["mcpar_read_order"] = order
    facts["mcpar_read_line"] = io_raw[:io_raw.index("void readMcPar(")].count("\n") + 1
    with open(os.path.join(HERE, "ref_facts.json"), "w") as f:
        json.dump(facts, f, indent=1)
    print(json.dumps(facts, indent=1)[:3000])


if __name__ == "__main__":
    main()
