"""Regenerates tests/golden/mc_proc_listing.json:  python tests/golden/make_h5_listing.py  (needs /root/reference)

The layout of mc_proc_<rank>.h5 as the reference's printPhotons writes it, read off its HDF5 call sequence in
Src/mcrat_io.c -- the reference cannot be run here (GSL is absent), but what it asks of the HDF5 library is in its text: the file and
group name formats, and for every H5Dcreate2 inside printPhotons the dataset name, the memory type, the dataspace / creation-property
variables and the compile-time switch it sits under; the chunk and maximum dimensions those variables are made with; and the append
idiom (H5Dset_extent to old + new, hyperslab at offset old).  Only these facts are stored -- no source text."""
import json
import os
import re
import sys

REF = "/root/reference/Src/mcrat_io.c"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read()
    start = src.index("void printPhotons(")
    end = src.index("\nint saveCheckpoint(")
    body = src[start:end]
    first_line = src[:start].count("\n") + 1
    # the switch every line sits under (#if X == ON ... #endif, not nested in this function)
    switch_at, cur = [], None
    for line in body.split("\n"):
        m = re.match(r"\s*#if\s+(\w+)\s*==\s*ON", line)
        if m:
            cur = m.group(1)
        elif re.match(r"\s*#endif", line):
            cur = None
        switch_at.append(cur)
    line_of = lambda pos: body[:pos].count("\n")
    datasets, seen = [], set()
    for m in re.finditer(r'H5Dcreate2\s*\(\s*group_id\s*,\s*"(\w+)"\s*,\s*(H5T_NATIVE_\w+)\s*,\s*(\w+)\s*,\s*H5P_DEFAULT\s*,\s*(\w+)\s*,', body):
        name, mem_type, space, prop = m.groups()
        if name in seen:
            continue
        seen.add(name)
        datasets.append({"name": name, "memory_type": mem_type, "switch": switch_at[line_of(m.start())], "dataspace": space, "dcpl": prop,
                         "line": first_line + line_of(m.start())})
    spaces = {}
    for m in re.finditer(r"(\w+)\s*=\s*H5Screate_simple\s*\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*\)", body):
        if m.group(4) != "NULL":
            spaces[m.group(1)] = {"rank_var": m.group(2), "dims_var": m.group(3), "maxdims_var": m.group(4)}
    chunks = {m.group(1): {"rank_var": m.group(2), "dims_var": m.group(3)} for m in re.finditer(r"H5Pset_chunk\s*\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*\)", body)}
    decl = {
        "rank": int(re.search(r"\brank\s*=\s*(\d+)", body).group(1)),
        "maxdims": re.search(r"maxdims\[1\]\s*=\s*\{\s*(\w+)\s*\}", body).group(1),
        "dims": re.search(r"\bdims\[1\]\s*=\s*\{\s*(\w+)\s*\}", body).group(1),
        "dims_weight": re.search(r"dims_weight\[1\]\s*=\s*\{\s*(\w+)\s*\}", body).group(1),
    }
    names = re.search(r'snprintf\(mc_file,sizeof\(mc_file\),"([^"]+)",dir,"([^"]+)",\s*angle_rank,\s*"([^"]+)"', body)
    group = re.search(r'snprintf\(group,sizeof\(group\),"([^"]+)",frame', body)
    append = {
        "extends": len(re.findall(r"H5Dset_extent\s*\(", body)),
        "size_is_old_plus_new": bool(re.search(r"size\[0\]\s*=\s*dims\[0\]\s*\+\s*dims_old\[0\]", body)),
        "hyperslab_offset_is_old": bool(re.search(r"offset\[0\]\s*=\s*dims_old\[0\]", body)),
        "hyperslab": re.search(r"H5Sselect_hyperslab\s*\(\s*fspace\s*,\s*(\w+)\s*,\s*offset\s*,\s*NULL\s*,\s*\n?\s*dims\s*,\s*NULL\s*\)", body).group(1),
        "group_exists_test": "H5Gget_objinfo" if "H5Gget_objinfo (file, group" in body else None,
    }
    out = {"source": "Src/mcrat_io.c printPhotons, lines %d-%d" % (first_line, first_line + body.count("\n")),
           "file_name": {"format": names.group(1), "prefix": names.group(2), "suffix": names.group(3)},
           "group_name": {"format": group.group(1), "of": "frame"},
           "declarations": decl, "dataspaces": spaces, "chunking": chunks, "datasets": datasets, "append": append}
    with open(os.path.join(HERE, "mc_proc_listing.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("%d datasets:" % len(datasets), " ".join(d["name"] + ("[%s]" % d["switch"] if d["switch"] else "") for d in datasets))


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("the reference is not here: the committed mc_proc_listing.json stands")
    main()
