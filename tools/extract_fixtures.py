#!/usr/bin/env python3
"""Extract DATA fixtures (names, numbers) from the reference's generated data files.

Runs only in the build container (reads /root/reference as text); its outputs under
tests/golden/ are what travels to the GPU box.  Nothing here executes reference code and
no reference source text is kept -- only the facts (instance names, label values,
quantities, prices, ENI limits) those data files hold.

Sources (relative to /root/reference):
  website/content/en/preview/reference/instance-types.md     776 known-answer rows
      (produced by hack/docs/instancetypes_gen/main.go:88-233: AL2023, no block device
      mappings, VMMemoryOverheadPercent=0.075; labels + Capacity - Overhead.Total())
  pkg/providers/pricing/zz_generated.pricing_aws.go          on-demand price table
  pkg/providers/instancetype/zz_generated.vpclimits.go       ENI / IP limits
  pkg/providers/instancetype/zz_generated.bandwidth.go       network bandwidth labels
  pkg/fake/zz_generated.describe_instance_types.go           16-type EC2 fixture
  pkg/fake/ec2api.go:555-674                                 zonal offerings of the fixture
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def doc_rows():
    txt = read("website/content/en/preview/reference/instance-types.md")
    rows, cur, mode = [], None, None
    for line in txt.splitlines():
        line = line.strip()
        m = re.match(r"^### `(.+)`$", line)
        if m:
            cur = {"name": m.group(1), "labels": {}, "resources": {}}
            rows.append(cur)
            mode = None
            continue
        if line == "#### Labels":
            mode = "labels"
            continue
        if line == "#### Resources":
            mode = "resources"
            continue
        if cur is None or mode is None or not line.startswith("|"):
            continue
        cells = line.strip("|").split("|")
        if len(cells) != 2 or cells[0] in ("Label", "Resource", "--", " Label ", " Resource "):
            continue
        k, v = cells[0].strip(), cells[1].strip()
        if k in ("Label", "Resource", "--"):
            continue
        cur[mode][k] = v
    return rows


def pricing():
    txt = read("pkg/providers/pricing/zz_generated.pricing_aws.go")
    gen = re.search(r"generated at (\S+) for (\S+)", txt)
    body = txt[txt.index('"us-east-1": {'):]
    prices = {m.group(1): float(m.group(2)) for m in re.finditer(r'"([a-z0-9\-]+\.[a-z0-9\-]+)":\s*([0-9.]+)', body)}
    return {"generated": gen.group(1), "region": gen.group(2), "on_demand": prices}


def vpclimits():
    txt = read("pkg/providers/instancetype/zz_generated.vpclimits.go")
    out = {}
    for m in re.finditer(r'\n\t"([a-z0-9\-]+\.[a-z0-9\-]+)": \{(.*?)\n\t\},', txt, re.S):
        name, body = m.group(1), m.group(2)

        def g(field, conv=int):
            mm = re.search(field + r":\s*([^,\n]+),", body)
            return conv(mm.group(1).strip().strip('"')) if mm else None
        cards = [[int(a), int(b)] for a, b in re.findall(
            r"MaximumNetworkInterfaces:\s*(\d+),\s*NetworkCardIndex:\s*(\d+),", body)]
        out[name] = {
            "interface": g("Interface"), "ipv4_per_interface": g("IPv4PerInterface"),
            "trunking": g("IsTrunkingCompatible", lambda s: s == "true"),
            "branch_interface": g("BranchInterface"),
            "default_card": g("DefaultNetworkCardIndex"),
            "cards": cards,  # [max interfaces, card index]
            "hypervisor": g("Hypervisor", str), "bare_metal": g("IsBareMetal", lambda s: s == "true"),
        }
    return out


def bandwidth():
    txt = read("pkg/providers/instancetype/zz_generated.bandwidth.go")
    return {m.group(1): int(m.group(2)) for m in re.finditer(r'"([a-z0-9\-]+\.[a-z0-9\-]+)":\s*(\d+),', txt)}


# ---- tiny reader for the Go composite literal that holds the 16-type EC2 fixture ----
class GoLit:
    def __init__(self, s):
        self.s, self.i = s, 0

    def ws(self):
        while self.i < len(self.s):
            if self.s[self.i].isspace() or self.s[self.i] == ",":
                self.i += 1
            elif self.s.startswith("//", self.i):
                self.i = self.s.index("\n", self.i)
            else:
                break

    def value(self):
        self.ws()
        s = self.s
        m = re.compile(r'aws\.(String|Int64|Bool|Float64)\(([^()]*)\)').match(s, self.i)
        if m:
            self.i = m.end()
            kind, raw = m.group(1), m.group(2).strip()
            if kind == "String":
                return raw.strip('"')
            if kind == "Int64":
                return int(raw)
            if kind == "Bool":
                return raw == "true"
            return float(raw)
        m = re.compile(r'aws\.StringSlice\(\[\]string\{([^}]*)\}\)').match(s, self.i)
        if m:
            self.i = m.end()
            return re.findall(r'"([^"]*)"', m.group(1))
        m = re.compile(r'\[\]\*?[\w.]+\{').match(s, self.i)
        if m:  # slice of structs
            self.i = m.end()
            items = []
            while True:
                self.ws()
                if s[self.i] == "}":
                    self.i += 1
                    return items
                if s[self.i] == "{":
                    self.i += 1
                    items.append(self.fields())
                else:
                    items.append(self.value())
        m = re.compile(r'&?[\w.]+\{').match(s, self.i)
        if m:
            self.i = m.end()
            return self.fields()
        raise ValueError("unparsed at: " + s[self.i:self.i + 60])

    def fields(self):
        d = {}
        while True:
            self.ws()
            if self.s[self.i] == "}":
                self.i += 1
                return d
            m = re.compile(r'(\w+):').match(self.s, self.i)
            if not m:
                raise ValueError("field expected at: " + self.s[self.i:self.i + 60])
            self.i = m.end()
            d[m.group(1)] = self.value()


def fake16():
    txt = read("pkg/fake/zz_generated.describe_instance_types.go")
    start = txt.index("InstanceTypes: []*ec2.InstanceTypeInfo{")
    p = GoLit(txt)
    p.i = start + len("InstanceTypes: ")
    infos = p.value()
    api = read("pkg/fake/ec2api.go")
    offs = re.findall(r'InstanceType:\s*aws\.String\("([^"]+)"\),\s*Location:\s*aws\.String\("([^"]+)"\)', api)
    zones = {}
    for it, z in offs:
        zones.setdefault(it, [])
        if z not in zones[it]:
            zones[it].append(z)
    subnets = [{"zone": z, "zone_id": zid} for z, zid in re.findall(
        r'AvailabilityZone:\s*aws\.String\("([^"]+)"\),\s*AvailabilityZoneId:\s*aws\.String\("([^"]+)"\)', api)]
    return {"instance_types": infos, "offering_zones": zones, "subnets": subnets}


def main():
    os.makedirs(OUT, exist_ok=True)
    rows = doc_rows()
    assert len(rows) == 776, len(rows)

    def dump(name, obj):
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=None, separators=(",", ":"), sort_keys=True)
            f.write("\n")
    dump("instance_types_doc.json", rows)
    pr = pricing()
    dump("pricing_od.json", pr)
    vl = vpclimits()
    dump("vpclimits.json", vl)
    bw = bandwidth()
    dump("bandwidth.json", bw)
    fk = fake16()
    dump("fake_ec2_16.json", fk)
    print("doc rows", len(rows), "prices", len(pr["on_demand"]), "vpclimits", len(vl),
          "bandwidth", len(bw), "fake types", len(fk["instance_types"]))


if __name__ == "__main__":
    main()
