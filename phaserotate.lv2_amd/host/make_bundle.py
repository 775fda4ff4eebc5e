#!/usr/bin/env python3
"""Assemble an LV2 bundle directory around the MI355X plugin binary:

    <out>/phaserotate.lv2/{manifest.ttl, phaserotate.ttl, phaserotate.so, libprot_hip.so}

The Turtle is generated from the port table below; URIs, port indices, symbols
and ranges are the interface facts of the reference bundle (lv2ttl/*.in,
src/phaserotate.h:99-111) so that sessions saved with x42 phaserotate.lv2 load
unchanged.  No GUI (robtk) and no JACK wrapper: out of scope.

usage: make_bundle.py <out-dir>
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
URI = "http://gareus.org/oss/lv2/phaserotate"

PREFIXES = """@prefix atom:  <http://lv2plug.in/ns/ext/atom#> .
@prefix doap:  <http://usefulinc.com/ns/doap#> .
@prefix lv2:   <http://lv2plug.in/ns/lv2core#> .
@prefix pprop: <http://lv2plug.in/ns/ext/port-props#> .
@prefix rdf:   <http://www.w3.org/1999/02/22-rdf-syntax-ns#> .
@prefix rdfs:  <http://www.w3.org/2000/01/rdf-schema#> .
@prefix rsz:   <http://lv2plug.in/ns/ext/resize-port#> .
@prefix units: <http://lv2plug.in/ns/extensions/units#> .
@prefix urid:  <http://lv2plug.in/ns/ext/urid#> .
"""


def atom_port(index, symbol, name, direction, extra=""):
    return ("[ a atom:AtomPort, lv2:%sPort ; atom:bufferType atom:Sequence ; lv2:designation lv2:control ;\n"
            "    lv2:index %d ; lv2:symbol \"%s\" ; lv2:name \"%s\" ;%s ]" % (direction, index, symbol, name, extra))


def angle_port(index, symbol, name):
    pts = " ".join("lv2:scalePoint [ rdfs:label \"%d\\u00b0\" ; rdf:value %.1f ] ;" % (d, d)
                   for d in range(-180, 181, 45))
    return ("[ a lv2:InputPort, lv2:ControlPort ; lv2:index %d ; lv2:symbol \"%s\" ; lv2:name \"%s\" ;\n"
            "    lv2:default 0.0 ; lv2:minimum -180.0 ; lv2:maximum 180.0 ; units:unit units:degree ;\n"
            "    pprop:rangeSteps 721 ; %s ]" % (index, symbol, name, pts))


def audio_port(index, symbol, name, direction):
    return "[ a lv2:AudioPort, lv2:%sPort ; lv2:index %d ; lv2:symbol \"%s\" ; lv2:name \"%s\" ]" % (
        direction, index, symbol, name)


def plugin(suffix, label, channels, deferred=0):
    ports = [
        atom_port(0, "control", "control", "Input"),
        atom_port(1, "notify", "Notify", "Output", " rsz:minimumSize 4096 ;"),
        "[ a lv2:ControlPort, lv2:OutputPort ; lv2:index 2 ; lv2:symbol \"latency\" ; lv2:name \"Signal Latency\" ;\n"
        "    lv2:minimum 0 ; lv2:maximum %d ; lv2:portProperty lv2:reportsLatency, lv2:integer ; units:unit units:frame ]" % (16384 if deferred else 8192),
    ]
    for c, (sfx, nm) in enumerate(channels):
        ports.append(angle_port(3 + 3 * c, "angle" + sfx, ("Angle " + nm).strip()))
        ports.append(audio_port(4 + 3 * c, "in" + sfx, ("In " + nm).strip(), "Input"))
        ports.append(audio_port(5 + 3 * c, "out" + sfx, ("Out " + nm).strip(), "Output"))
    return (("<%s%s>\n  a lv2:Plugin, doap:Project, lv2:UtilityPlugin ;\n"
            "  doap:license <http://usefulinc.com/doap/licenses/gpl> ;\n"
            "  doap:name \"Phaserotate%s (MI355X)\" ;\n"
            # (no lv2:hardRTCapable by default, unlike lv2ttl/phaserotate.ports.in:7: run() waits for the device -- it never
            #  allocates, but its duration is a host-link round trip, not a bounded CPU computation.  --deferred N: run() posts its
            #  period and returns the one before without waiting, N frames more latency: declared again)
            + ("  lv2:optionalFeature lv2:hardRTCapable ;\n" if deferred else "") +
            "  lv2:optionalFeature <http://lv2plug.in/ns/ext/options#options> ;\n  lv2:requiredFeature urid:map ;\n"
            "  rdfs:comment \"Phase rotate signal; DSP on an AMD MI355X through libprot_hip.\" ;\n"
            "  lv2:port %s .\n") % (URI, suffix, label, " ,\n  ".join(ports)))


def main(out, deferred=0):
    """deferred = N > 0: the bundle asks the plugin for its deferred run() (include/prot_hip.h, prot_lv2_set_deferred) with N frames
    of extra latency (1 = the host's maxBlockLength) through a file `deferred` beside the binary, and declares lv2:hardRTCapable"""
    bundle = os.path.join(out, "phaserotate.lv2")
    os.makedirs(bundle, exist_ok=True)
    with open(os.path.join(bundle, "manifest.ttl"), "w") as f:
        f.write("@prefix lv2:  <http://lv2plug.in/ns/lv2core#> .\n@prefix rdfs: <http://www.w3.org/2000/01/rdf-schema#> .\n\n")
        for sfx in ("", "#stereo"):
            f.write("<%s%s> a lv2:Plugin ; lv2:binary <phaserotate.so> ; rdfs:seeAlso <phaserotate.ttl> .\n" % (URI, sfx))
    with open(os.path.join(bundle, "phaserotate.ttl"), "w") as f:
        f.write(PREFIXES + "\n")
        f.write(plugin("", "", [("", "")], deferred) + "\n")
        f.write(plugin("#stereo", " Stereo", [("L", "Left"), ("R", "Right")], deferred))
    dpath = os.path.join(bundle, "deferred")
    if deferred:
        with open(dpath, "w") as f:
            f.write("%d\n" % deferred)
    elif os.path.exists(dpath):
        os.remove(dpath)
    for lib in ("phaserotate.so", "libprot_hip.so"):
        src = os.path.join(LIBDIR, lib)
        if os.path.exists(src):
            shutil.copy(src, bundle)  # phaserotate.so finds libprot_hip.so through its $ORIGIN rpath
    print(bundle)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out", nargs="?", default=".")
    ap.add_argument("--deferred", type=int, nargs="?", const=1, default=0,
                    help="frames of extra latency of the deferred run() (1 = the host's maxBlockLength); declares lv2:hardRTCapable")
    a = ap.parse_args()
    main(a.out, a.deferred)
