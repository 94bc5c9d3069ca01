#!/usr/bin/env python3
"""Applies the host <-> device synchronisation edits of INTEGRATION.md section 3 to the reference main program.

    patch_main.py <reference src/q-gcm.F> <output file>

The output is a patched COPY of the reference's main program: a build product, written only into the ignored
build directory of build_dropin.sh (never committed, deleted after it has been compiled).  Everything outside the
main program (the subroutines that follow it in q-gcm.F) is passed through unchanged.  The edits:

  E1  USE vorsubs                  -> USE vorsubs, ONLY : qcomp, merqcy  + USE vorsubs_hip [+ vorsubs_hip_at]
      (ocqbdy / atqzbd come from the shim; it also serves the start-up calls on host arrays)
  E2  after USE valsubs            -> USE qgcm_hip_iface, qgcm_hip_state, valsubs_hip [, qgcm_hip_atstate]
  E3  after "call homsol"          -> call qgcm_hip_push [+ qgcm_hip_oml_push] [+ qgcm_hip_atm_push]
  E4  "call valids (solnok)" inside the time loop -> device scan first, pull + the reference's valids on failure
  E5  before every diagnostic / output call after the loop starts -> pull the device state
  E6  leapfrog averaging blocks    -> qgcm_hip_lf_average on the device handles (ast / hmixa stay on the host)
  E7  coupled / atmos builds: pull before xforc / aml, push the forcing after them
  E8  before the final "stop"      -> shut the handles down
"""
import re
import sys

PULL_OC = ["#ifndef atmos_only", "      call qgcm_hip_pull", "#  ifndef no_oml_k247", "      call qgcm_hip_oml_pull",
           "#  endif", "#endif"]
PULL_AT = ["#ifndef ocean_only", "      call qgcm_hip_atm_pull", "#endif"]
PULL = PULL_OC + PULL_AT
DIAG = ("monnc_comp", "prsamp", "cfltry", "areavg", "resave", "resave_nc", "ocnc_out", "atnc_out", "qocdiag_out",
        "tavocn", "tavatm", "covocn", "covatm", "constr", "ocnc_avgout_k247", "monnc_out")


def patch(lines):
    out = []
    n = len(lines)
    end_main = next(i for i, l in enumerate(lines) if re.match(r"\s+SUBROUTINE ipbget", l))
    i_loop = next(i for i, l in enumerate(lines) if re.match(r"\s+do 1000 nt=", l))
    done = set()
    i = 0
    while i < n:
        l = lines[i]
        if i >= end_main:
            out.append(l)
            i += 1
            continue
        s = l.strip()
        if re.fullmatch(r"USE vorsubs", s) and "E1" not in done:
            out += ["      USE vorsubs, ONLY : qcomp, merqcy", "#ifndef atmos_only", "      USE vorsubs_hip", "#endif",
                    "#ifndef ocean_only", "      USE vorsubs_hip_at", "#endif"]
            done.add("E1")
        elif re.fullmatch(r"USE valsubs", s) and "E2" not in done:
            out += [l, "      USE qgcm_hip_iface", "#ifndef atmos_only", "      USE qgcm_hip_state", "      USE valsubs_hip",
                    "#endif", "#ifndef ocean_only", "      USE qgcm_hip_atstate", "#endif"]
            done.add("E2")
        elif re.fullmatch(r"call homsol", s) and "E3" not in done:
            out += [l, "#ifndef atmos_only", "      call qgcm_hip_push", "#  ifndef no_oml_k247", "      call qgcm_hip_oml_push",
                    "#  endif", "#endif", "#ifndef ocean_only", "      call qgcm_hip_atm_push", "#endif"]
            done.add("E3")
        elif i > i_loop and re.fullmatch(r"call valids \(solnok\)", s):
            ind = l[:len(l) - len(l.lstrip())]
            out += ["#ifdef atmos_only"] + PULL_AT + [l, "#else",
                    ind + "call valids_hip (solnok)",
                    ind + "if ( .not.solnok ) then"] + PULL + [
                    ind + "  solnok = .true.",
                    ind + "  call valids (solnok)",
                    ind + "endif", "#endif"]
            done.add("E4")
        elif i > i_loop and re.match(r"\s+if \( mod\(nt-1,25\*nstr\)\.eq\.0 \) then", l):
            # E6 ocean: replace the body up to the matching endif (the one before "#endif /* atmos_only */")
            j = i + 1
            while not lines[j].startswith("#endif /* atmos_only */"):
                j += 1
            k = j - 1
            while lines[k].strip() != "endif":
                k -= 1
            out += [l, "          call qgcm_hip_check(qgcm_hip_lf_average(qgcm_hip_handle), 'lf_average')", lines[k]]
            i = k + 1
            done.add("E6o")
            continue
        elif i > i_loop and re.match(r"\s+if \( mod\(nt-1,100\)\.eq\.0 \) then", l):
            j = i + 1
            depth = 1
            while depth:
                t = lines[j].strip()
                if re.match(r"if .* then$", t):
                    depth += 1
                elif t == "endif":
                    depth -= 1
                j += 1
            out += [l, "          call qgcm_hip_check(qgcm_hip_lf_average(qgcm_hip_atm_handle), 'lf_average (atmosphere)')",
                    "*         mixed-layer fields of aml stay with the host (src/q-gcm.F:1388-1394)",
                    "          ast = 0.5d0*( ast + astm )", "          hmixa = 0.5d0*( hmixa + hmixam )", lines[j - 1]]
            i = j
            done.add("E6a")
            continue
        elif i > i_loop and re.fullmatch(r"call (xforc|aml)", s):
            out += PULL + [l, "#ifndef atmos_only", "      call qgcm_hip_push_forcing", "#  ifndef no_oml_k247",
                           "      call qgcm_hip_oml_push", "#  endif", "#endif",
                           "#ifndef ocean_only", "      call qgcm_hip_atm_push_forcing", "#endif"]
            done.add("E7")
        elif i > i_loop and (m := re.match(r"(\s+)if (\(.*\)) call (\w+)\s*$", l)) and m.group(3) in DIAG:
            ind = m.group(1)
            out += [ind + "if " + m.group(2) + " then"] + PULL + [ind + "  call " + m.group(3), ind + "endif"]
            done.add("E5")
        elif i > i_loop and (m := re.match(r"\s+call (\w+)", l)) and m.group(1) in DIAG:
            out += PULL + [l]
            done.add("E5")
        elif i > i_loop and s == "stop" and lines[i + 1].strip() == "end" and "E8" not in done:
            out += ["#ifndef atmos_only", "      call qgcm_hip_shutdown", "#endif", "#ifndef ocean_only",
                    "      call qgcm_hip_atm_shutdown", "#endif", l]
            done.add("E8")
        else:
            out.append(l)
        i += 1
    need = {"E1", "E2", "E3", "E4", "E5", "E6o", "E6a", "E8"}
    missing = need - done
    if missing:
        raise SystemExit("patch_main.py: edit points not found in the reference main program: %s" % sorted(missing))
    return out


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    text = open(src).read().split("\n")
    open(dst, "w").write("\n".join(patch(text)))
