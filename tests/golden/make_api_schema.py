#!/usr/bin/env python3
"""Generates tests/golden/api_schema.json from the reference's FlatBuffers header core/api_generated.h -- flatc's own output for
core/api.fbs, i.e. the wire contract a reader built from the reference uses: for every table its vtable slots with type, default
and alignment as the generated accessors / verifiers state them; for every struct its size, alignment and member offsets; the
enums and the members of the PopModel union.  Data only (names and numbers), and produced by this script, not by hand: the
.dphy writer (delphy_amd/csrc/emat_dphy.cpp) is checked against it by tests/test_dphy_writer.py without sharing a table with it.
Run in the build container (where /root/reference exists); the JSON is committed."""
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/core/api_generated.h"
text = open(SRC).read()
SCALARS = {"int8_t": ("b", 1), "uint8_t": ("B", 1), "int16_t": ("h", 2), "uint16_t": ("H", 2), "int32_t": ("i", 4), "uint32_t": ("I", 4),
           "int64_t": ("q", 8), "uint64_t": ("Q", 8), "float": ("f", 4), "double": ("d", 8)}

enums = {}
for m in re.finditer(r"enum (\w+) : (\w+) \{(.*?)\};", text, re.S):
    name, base, body = m.groups()
    vals = {}
    for k, v in re.findall(r"%s_(\w+) = (-?\d+)" % name, body):
        if k not in ("MIN", "MAX"):
            vals[k] = int(v)
    enums[name] = {"base": base, "values": vals}

structs = {}
for m in re.finditer(r"FLATBUFFERS_MANUALLY_ALIGNED_STRUCT\((\d+)\) (\w+) FLATBUFFERS_FINAL_CLASS \{\s*private:(.*?)public:", text, re.S):
    align, name, body = int(m.group(1)), m.group(2), m.group(3)
    off, fields = 0, []
    for ctype, member in re.findall(r"^\s*(\w+) (\w+)_;", body, re.M):
        fmt, size = SCALARS[ctype]
        off = (off + size - 1) // size * size
        fields.append({"name": member.rstrip("_"), "ctype": ctype, "fmt": fmt, "offset": off, "size": size, "padding": member.startswith("padding")})
        off += size
    size = int(re.search(r"FLATBUFFERS_STRUCT_END\(%s, (\d+)\)" % name, text).group(1))
    assert (off + align - 1) // align * align == size, (name, off, size)
    structs[name] = {"align": align, "size": size, "fields": fields}

tables = {}
for m in re.finditer(r"struct (\w+) FLATBUFFERS_FINAL_CLASS : private ::flatbuffers::Table \{(.*?)\n\};", text, re.S):
    name, body = m.groups()
    slots = {k.lower(): int(v) for k, v in re.findall(r"VT_(\w+) = (\d+)", body)}
    fields = {}
    for ctype, vt, default in re.findall(r"GetField<(\w+)>\(VT_(\w+), ([^)]*)\)", body):
        fmt, size = SCALARS[ctype]
        d = default.strip().rstrip("fL")
        d = d[:-1] if d.endswith("L") else d
        fields[vt.lower()] = {"kind": "scalar", "ctype": ctype, "fmt": fmt, "size": size, "default": float(d) if ctype in ("float", "double") else int(d)}
    for target, vt in re.findall(r"GetPointer<const (.+?) \*>\(VT_(\w+)\)", body):
        target = target.strip()
        if target == "::flatbuffers::String":
            f = {"kind": "string"}
        elif target == "void":
            f = {"kind": "union"}
        elif target.startswith("::flatbuffers::Vector<"):
            inner = target[len("::flatbuffers::Vector<"):-1].strip()
            mo = re.match(r"::flatbuffers::Offset<delphy::api::(\w+)>", inner)
            ms = re.match(r"const delphy::api::(\w+) \*", inner)
            if mo:
                f = {"kind": "vector_of_tables", "table": mo.group(1)}
            elif ms:
                f = {"kind": "vector_of_structs", "struct": ms.group(1)}
            else:
                f = {"kind": "vector_of_scalars", "ctype": inner, "fmt": SCALARS[inner][0], "size": SCALARS[inner][1]}
        else:
            f = {"kind": "table", "table": re.match(r"delphy::api::(\w+)", target).group(1)}
        fields.setdefault(vt.lower(), f)
    for ctype, vt, align in re.findall(r"VerifyField<(\w+)>\(verifier, VT_(\w+), (\d+)\)", body):
        assert fields[vt.lower()]["ctype"] == ctype
        fields[vt.lower()]["align"] = int(align)
    for vt in re.findall(r"VerifyOffset\(verifier, VT_(\w+)\)", body):
        assert fields[vt.lower()]["kind"] != "scalar"
    for k in fields:
        fields[k]["vt"] = slots[k]
    # a union value's discriminator is the field "<name>_type" flatc adds in the slot before it
    for k, f in fields.items():
        if f["kind"] == "union":
            f["type_field"] = k + "_type"
            assert fields[k + "_type"]["vt"] == f["vt"] - 2
    assert set(fields) == set(slots), (name, set(slots) - set(fields))
    tables[name] = {"fields": fields}

unions = {"PopModel": {k: v for k, v in enums["PopModel"]["values"].items() if k != "NONE"}}
for t in tables.values():
    for k, f in t["fields"].items():
        if f["kind"] == "union":
            f["union"] = "PopModel"
out = {"source": "reference core/api_generated.h (flatc output for core/api.fbs)", "enums": enums, "structs": structs, "tables": tables, "unions": unions}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api_schema.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, "-", len(tables), "tables,", len(structs), "structs,", sum(len(t["fields"]) for t in tables.values()), "table fields")
