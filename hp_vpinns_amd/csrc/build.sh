#!/bin/bash
# Build libhpvpinn.so for gfx950 in-tree (hipcc cross-compiles without a GPU), every stale object in parallel.
#   ../libhpvpinn.so            the product
#   ../libhpvpinn_testhooks.so  the same sources with -DHPV_TEST_HOOKS -DHPV_EXPERIMENTS: the fault-injection knobs of the tests
#                               (HPV_DEBUG_SPLIT_SKIP: a partner workgroup stays away from an in-kernel exchange) and every
#                               measured-slower kernel variant kept as evidence (HPV_PERSIST, HPV_PJ_PIPE / _STREAM / _DMA / _GRID /
#                               _OCC_PAD, HPV_FUSED_GSTASH, HPV_WIDE_RC, HPV_TILE_DEBUG) exist ONLY there -- the product library
#                               neither reads those variables nor carries the kernels (tests/test_cabi.py greps the binary)
# From scratch on 8 cores: ~85 s (round 4: 150 s).  What it took: the experiments out of the product objects, one compilation
# instead of two for the AGPR-guarded files (the guard reads the assembly -save-temps leaves behind), the width-generic kernels as
# one translation unit per width AND input dimension (the 64-wide unit alone was 94 s), longest jobs started first.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $HPV_EXTRA_FLAGS"   # e.g. HPV_EXTRA_FLAGS=-DHPV_FZ_TIMING
SRCS="kernels_mfma kernels_fused kernels_fused_gen kernels_project kernels_tile kernels_tall kernels_generic hpv_api hpv_exchange hpv_bench"
ELEM_SHAPES="20,20,10,10 16,16,8,8 12,12,6,6"             # kernels_elem.hip: one object per element shape (= HPV_ELEM_SHAPES of hpv_mfma_dev.h)
WIDE_WIDTHS="64 32 48 24 40"                            # kernels_wide.hip: one object per hidden width and dimension (= HPV_WIDE_WIDTHS of hpv_mfma.h)
HOOKED="kernels_fused kernels_project kernels_tile kernels_tall hpv_api hpv_exchange"   # the sources that contain test hooks / experiments (built twice)
WIDE_HOOKED="24_d2"                                     # width_dimension units whose experiment (HPV_WIDE_RC) is built into the test-hooks library
TH_FLAGS="-DHPV_TEST_HOOKS -DHPV_EXPERIMENTS"
CHK="python3 ../../scripts/check_agpr.py"
# objects are cached by mtime; a change of flags must invalidate them (.flags remembers what the objects were built with)
if [ "$(cat .flags 2>/dev/null)" != "$FLAGS|$HPV_FUSED_EXTRA|v2" ]; then rm -f *.o; echo "$FLAGS|$HPV_FUSED_EXTRA|v2" > .flags; fi

stale() {   # stale <object> <source>: the object is missing or older than its source / any header
  [ ! -f "$1" ] && return 0
  local d
  [ "$2" = kernels_fused_gen.hip ] && [ kernels_fused.hip -nt "$1" ] && return 0      # (that unit IS kernels_fused.hip, its other instantiations)
  for d in "$2" hpv_ctx.h hpv_internal.h hpv_mfma.h hpv_mfma_dev.h hpv_wide_dev.h hpv_project_wg.h hpv_math.h hpv_fused_dev.h ../../include/hpvpinn.h; do
    [ -f "$d" ] && [ "$d" -nt "$1" ] && return 0
  done
  return 1
}

# guard <asm> <mangled-name fragment> <hand-managed base>...: 0 clear, 1 tripped (some instantiation overlaps), 2 check impossible
guard() {
  local asm=$1 rc=0; shift
  while [ $# -ge 2 ]; do
    local r=0; $CHK $asm $1 $2 >&2 || r=$?
    [ $r -eq 3 ] && { echo "build.sh: ERROR -- $1 spills registers to scratch memory (scripts/check_agpr.py): a hand-scheduled kernel must not" >&2; return 2; }
    [ $r -ge 2 ] && return 2
    [ $r -eq 1 ] && rc=1
    shift 2
  done
  return $rc
}

# Two sources park live values in hand-chosen AGPRs.  They are compiled ONCE with -save-temps: scripts/check_agpr.py verifies on the
# device assembly the object was made from, per template instantiation, that the compiler's own registers stay clear of the
# hand-managed range.  If a compiler release ever needs more (exit 1), the file is compiled again WITHOUT that kernel / instantiation
# (-DHPV_AGPR_GUARD_TRIPPED[_QT]: the launch functions decline, the callers fall back; hpv_build_info() reports it, bench.py
# prints it) -- and the build says so loudly.  If the check cannot run at all (exit 2: symbol not found after a rename, no
# assembly) the build FAILS.
guarded_compile() {   # guarded_compile <source stem> <object> <extra flags> <file-only flags>
  local f=$1 obj=$2 extra=$3 XF=$4 asm=${2%.o}.s tmp=.tmp_${2%.o} g=0
  rm -rf $tmp; mkdir -p $tmp
  $HIPCC $FLAGS $XF $extra -save-temps=obj -c $f.hip -o $tmp/$f.o 2>$asm.err || { cat $asm.err >&2; rm -rf $tmp; return 1; }
  cp $tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s $asm || { echo "build.sh: ERROR -- no device assembly behind $f.hip" >&2; rm -rf $tmp; return 1; }
  local add=""
  if [ $f = kernels_fused ]; then
    # (instantiations: <L, SPLIT, QT, GS = false, element shape>: the GS = true ones (test-hooks library only) hand-manage no registers; the
    #  hand-managed range starts at 256 - (tiles per wave - 2) x 10 L registers; the quarter-tile one sits closest to it and has its own fallback)
    local S=ELi20ELi20ELi10ELi10E
    g=0; guard $asm k_iter_fusedILi3ELb0ELb0ELb0${S}Lb0E 106 k_iter_fusedILi3ELb1ELb0ELb0${S}Lb0E 106 k_iter_fusedILi2ELb0ELb0ELb0${S}Lb0E 156 k_iter_fusedILi2ELb1ELb0ELb0${S}Lb0E 156 || g=$?
    [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check $f.hip" >&2; rm -rf $tmp; return 1; }
    if [ $g -eq 1 ]; then
      echo "build.sh: WARNING -- AGPR guard tripped in $f.hip: building without k_iter_fused (fallback = HPV_FUSE=b structure)" >&2
      add="-DHPV_AGPR_GUARD_TRIPPED"
    else
      g=0; guard $asm k_iter_fusedILi3ELb0ELb1ELb0${S}Lb0E 106 k_iter_fusedILi2ELb0ELb1ELb0${S}Lb0E 156 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the quarter-tile instantiation of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in the quarter-tile instantiation of k_iter_fused: building with 7 / 6 / 6 / 6 whole tiles per wave" >&2
        add="-DHPV_AGPR_GUARD_TRIPPED_QT"
      fi
      # the other element shapes (FZ_SHAPES of kernels_fused.hip): 16x16 / 8x8 (5 tiles per wave), 12x12 / 6x6 (3)
      local S16=ELi16ELi16ELi8ELi8E S12=ELi12ELi12ELi6ELi6E
      g=0; guard $asm k_iter_fusedILi3ELb0ELb0ELb0${S16}Lb0E 166 k_iter_fusedILi3ELb1ELb0ELb0${S16}Lb0E 166 k_iter_fusedILi2ELb0ELb0ELb0${S16}Lb0E 196 k_iter_fusedILi2ELb1ELb0ELb0${S16}Lb0E 196 \
                       k_iter_fusedILi3ELb0ELb1ELb0${S16}Lb0E 166 k_iter_fusedILi2ELb0ELb1ELb0${S16}Lb0E 196 \
                       k_iter_fusedILi3ELb0ELb0ELb0${S12}Lb0E 226 k_iter_fusedILi3ELb1ELb0ELb0${S12}Lb0E 226 k_iter_fusedILi3ELb0ELb1ELb0${S12}Lb0E 226 \
                       k_iter_fusedILi2ELb0ELb0ELb0${S12}Lb0E 236 k_iter_fusedILi2ELb1ELb0ELb0${S12}Lb0E 236 k_iter_fusedILi2ELb0ELb1ELb0${S12}Lb0E 236 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the extra element shapes of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in an extra element shape of k_iter_fused: those shapes run on the other structures" >&2
        add="$add -DHPV_FZ_NO_EXTRA_SHAPES"
      fi
      # the element loop (MULTI = last template argument true; the keys above match both values of it through their common prefix --
      # here the MULTI instantiations alone, with their own fallback: one workgroup per element on every grid size)
      # (their hand-managed range starts at min(stash base, 256 - 2 x accumulators): the spilled sums come back through the top 90 (L = 3) / 66 (L = 2) AGPRs)
      g=0; guard $asm k_iter_fusedILi2ELb0ELb0ELb0${S}Lb1E 156 k_iter_fusedILi2ELb0ELb1ELb0${S}Lb1E 156 \
                       k_iter_fusedILi3ELb0ELb0ELb0${S16}Lb1E 166 k_iter_fusedILi3ELb0ELb1ELb0${S16}Lb1E 166 k_iter_fusedILi2ELb0ELb0ELb0${S16}Lb1E 190 k_iter_fusedILi2ELb0ELb1ELb0${S16}Lb1E 190 \
                       k_iter_fusedILi3ELb0ELb0ELb0${S12}Lb1E 166 k_iter_fusedILi3ELb0ELb1ELb0${S12}Lb1E 166 k_iter_fusedILi2ELb0ELb0ELb0${S12}Lb1E 190 k_iter_fusedILi2ELb0ELb1ELb0${S12}Lb1E 190 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the element-loop instantiations of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in an element-loop instantiation of k_iter_fused: grids larger than the chip keep one workgroup per element" >&2
        add="$add -DHPV_FZ_NO_MULTI"
      fi
    fi
  fi
  if [ $f = kernels_fused_gen ]; then
    # the general variational forms on k_iter_fused (template tail <.., MULTI = false, NT2, GEN = true>): same stash, same bases per shape
    # and depth.  A trip compiles out, in this order of preference: the quarter-tile instantiations (-DHPV_FZ_GEN_NO_QT), the
    # four-channel ones (-DHPV_FZ_GEN_NO_NT2), everything (-DHPV_FZ_GEN_TRIPPED: those forms run on the separate launches).
    local S=ELi20ELi20ELi10ELi10E S16=ELi16ELi16ELi8ELi8E S12=ELi12ELi12ELi6ELi6E
    local keys3="" keys3q="" keys4="" keys4q=""
    for Lb in "3 106 166 226" "2 156 196 236"; do
      set -- $Lb
      keys3="$keys3 k_iter_fusedILi${1}ELb0ELb0ELb0${S}Lb0ELi0ELb1E $2 k_iter_fusedILi${1}ELb1ELb0ELb0${S}Lb0ELi0ELb1E $2"
      keys3="$keys3 k_iter_fusedILi${1}ELb0ELb0ELb0${S16}Lb0ELi0ELb1E $3 k_iter_fusedILi${1}ELb1ELb0ELb0${S16}Lb0ELi0ELb1E $3"
      keys3="$keys3 k_iter_fusedILi${1}ELb0ELb0ELb0${S12}Lb0ELi0ELb1E $4 k_iter_fusedILi${1}ELb1ELb0ELb0${S12}Lb0ELi0ELb1E $4"
      keys3q="$keys3q k_iter_fusedILi${1}ELb0ELb1ELb0${S}Lb0ELi0ELb1E $2 k_iter_fusedILi${1}ELb0ELb1ELb0${S16}Lb0ELi0ELb1E $3 k_iter_fusedILi${1}ELb0ELb1ELb0${S12}Lb0ELi0ELb1E $4"
      # four channels park one more tile per wave in LDS: the stash is 2 L x 5 registers shorter
      keys4="$keys4 k_iter_fusedILi${1}ELb0ELb0ELb0${S16}Lb0ELi1ELb1E $(($3 + 10 * $1)) k_iter_fusedILi${1}ELb1ELb0ELb0${S16}Lb0ELi1ELb1E $(($3 + 10 * $1))"
      keys4="$keys4 k_iter_fusedILi${1}ELb0ELb0ELb0${S12}Lb0ELi1ELb1E $(($4 + 10 * $1)) k_iter_fusedILi${1}ELb1ELb0ELb0${S12}Lb0ELi1ELb1E $(($4 + 10 * $1))"
      keys4q="$keys4q k_iter_fusedILi${1}ELb0ELb1ELb0${S16}Lb0ELi1ELb1E $(($3 + 10 * $1))"
      [ $1 = 2 ] && keys4="$keys4 k_iter_fusedILi2ELb0ELb0ELb0${S}Lb0ELi1ELb1E $(($2 + 20)) k_iter_fusedILi2ELb1ELb0ELb0${S}Lb0ELi1ELb1E $(($2 + 20))"      # (20x20 points: two hidden layers only)
    done
    g=0; guard $asm $keys3 || g=$?
    [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check $f.hip" >&2; rm -rf $tmp; return 1; }
    if [ $g -eq 1 ]; then
      echo "build.sh: WARNING -- AGPR guard tripped in the general forms of k_iter_fused: those forms run on the separate launches" >&2
      add="-DHPV_FZ_GEN_TRIPPED"
    else
      g=0; guard $asm $keys4 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the four-channel instantiations of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in a four-channel instantiation of k_iter_fused: those forms run on the separate launches" >&2
        add="-DHPV_FZ_GEN_NO_NT2"; keys4q=""
      fi
      g=0; guard $asm $keys3q $keys4q || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the quarter-tile instantiations of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in a quarter-tile instantiation of the general forms: whole tiles only" >&2
        add="$add -DHPV_FZ_GEN_NO_QT"
      fi
      # the tight plan (FzPlan: four channels, three hidden layers, 20x20 points): twelve of the first stash place's fifteen doubles live in LDS
      if [ "$add" = "${add#*NO_NT2}" ]; then
        g=0; guard $asm k_iter_fusedILi3ELb0ELb0ELb0${S}Lb0ELi1ELb1E $((106 + 30 + 24)) k_iter_fusedILi3ELb1ELb0ELb0${S}Lb0ELi1ELb1E $((106 + 30 + 24)) || g=$?
        [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the tight-plan instantiation of $f.hip" >&2; rm -rf $tmp; return 1; }
        if [ $g -eq 1 ]; then
          echo "build.sh: WARNING -- AGPR guard tripped in the tight-plan instantiation of k_iter_fused: four channels on 20x20 points with three hidden layers run on the separate launches" >&2
          add="$add -DHPV_FZ_GEN_NO_TIGHT"
        fi
      else
        add="$add -DHPV_FZ_GEN_NO_TIGHT"
      fi
    fi
  fi
  if [ $f = kernels_tall ]; then    # same hand-managed AGPR stash (4 tiles x L x 5 doubles at the top of the file)
    # (template tail: <.., 80, 80, 5, 5, QT>; the QT instantiations keep one stash slot less: their range starts 30 registers higher)
    local T=ELi80ELi80ELi5ELi5
    g=0; guard $asm k_iter_tallILi2ELi1ELi3${T}ELb0 136 k_iter_tallILi2ELi0ELi3${T}ELb0 136 k_iter_tallILi2ELi1ELi2${T}ELb0 176 k_iter_tallILi2ELi0ELi2${T}ELb0 176 || g=$?
    [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check $f.hip" >&2; rm -rf $tmp; return 1; }
    if [ $g -eq 1 ]; then
      echo "build.sh: WARNING -- AGPR guard tripped in $f.hip: building without k_iter_tall (fallback = the separate launches)" >&2
      add="-DHPV_AGPR_GUARD_TRIPPED"
    else
      g=0; guard $asm k_iter_tallILi2ELi1ELi3${T}ELb1 166 k_iter_tallILi2ELi0ELi3${T}ELb1 166 k_iter_tallILi2ELi1ELi2${T}ELb1 196 k_iter_tallILi2ELi0ELi2${T}ELb1 196 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the quarter-tile instantiations of $f.hip" >&2; rm -rf $tmp; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in the quarter-tile instantiations of k_iter_tall: building with whole tiles only" >&2
        add="-DHPV_AGPR_GUARD_TRIPPED_QT"
      fi
    fi
  fi
  if [ -z "$add" ]; then mv $tmp/$f.o $obj; rm -rf $tmp; return 0; fi
  rm -rf $tmp
  $HIPCC $FLAGS $XF $add $extra -c $f.hip -o $obj      # (the guard tripped: once more, without the offending instantiation)
}

compile_one() {   # compile_one <source stem> <object> <extra flags>
  local f=$1 obj=$2 extra=$3
  if [ $f = kernels_fused ]; then guarded_compile $f $obj "$extra" "$HPV_FUSED_EXTRA"      # (A/B builds: flags for this file only, scripts/build_variant.sh --fused-only)
  elif [ $f = kernels_fused_gen ]; then guarded_compile $f $obj "$extra" "$HPV_FUSED_EXTRA"
  elif [ $f = kernels_tall ]; then guarded_compile $f $obj "$extra" ""
  else $HIPCC $FLAGS $extra -c $f.hip -o $obj; fi
}

# job pool: at most NJ compilers at a time (27 at once on 8 cores cost 30 % more CPU time than 8 at a time), longest jobs first;
# a job that fails leaves a marker (wait -n consumes exit statuses)
NJ=${HPV_BUILD_JOBS:-$(nproc)}
rm -f .fail_*
spawn() {   # spawn <object name> <command...>
  local name=$1; shift
  while [ "$(jobs -rp | wc -l)" -ge "$NJ" ]; do wait -n || true; done
  ( "$@" || { echo "build.sh: ERROR -- $name failed" >&2; rm -f $name; touch .fail_$name; } ) &
}
# the longest jobs first (they set the wall clock): the width-generic and the generic element-resident kernels
for w in $WIDE_WIDTHS; do
  for d in 2 1; do
    o=kernels_wide_${w}_d$d.o
    if stale $o kernels_wide.hip; then
      spawn $o $HIPCC $FLAGS -DHPV_WIDE_H=$w -DHPV_WIDE_D=$d -c kernels_wide.hip -o $o
    fi
  done
done
for sh in $ELEM_SHAPES; do
  IFS=, read qx qy ntx nty <<< "$sh"
  o=kernels_elem_${qx}_${qy}_${ntx}_${nty}.o
  if stale $o kernels_elem.hip; then
    spawn $o $HIPCC $FLAGS -DHPV_ELEM_QX=$qx -DHPV_ELEM_QY=$qy -DHPV_ELEM_NTX=$ntx -DHPV_ELEM_NTY=$nty -c kernels_elem.hip -o $o
  fi
done
for f in $SRCS; do
  [ -f $f.hip ] || continue
  if stale $f.o $f.hip; then spawn $f.o compile_one $f $f.o ""; fi
done
for f in $HOOKED; do
  if stale $f.th.o $f.hip; then spawn $f.th.o compile_one $f $f.th.o "$TH_FLAGS"; fi
done
for wd in $WIDE_HOOKED; do
  o=kernels_wide_$wd.th.o
  if stale $o kernels_wide.hip; then
    spawn $o $HIPCC $FLAGS -DHPV_WIDE_H=${wd%_d*} -DHPV_WIDE_D=${wd#*_d} $TH_FLAGS -c kernels_wide.hip -o $o
  fi
done
wait
if ls .fail_* >/dev/null 2>&1; then rm -f .fail_*; exit 1; fi

OBJS=""; TOBJS=""
for f in $SRCS; do
  [ -f $f.hip ] || continue
  OBJS="$OBJS $f.o"
  case " $HOOKED " in *" $f "*) TOBJS="$TOBJS $f.th.o";; *) TOBJS="$TOBJS $f.o";; esac
done
for w in $WIDE_WIDTHS; do
  for d in 1 2; do
    OBJS="$OBJS kernels_wide_${w}_d$d.o"
    case " $WIDE_HOOKED " in *" ${w}_d$d "*) TOBJS="$TOBJS kernels_wide_${w}_d$d.th.o";; *) TOBJS="$TOBJS kernels_wide_${w}_d$d.o";; esac
  done
done
for sh in $ELEM_SHAPES; do o=kernels_elem_${sh//,/_}.o; OBJS="$OBJS $o"; TOBJS="$TOBJS $o"; done
# -Bsymbolic: the two libraries may live in one process (the tests load both); each must bind its internal calls to itself
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../libhpvpinn.so $OBJS & l1=$!
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../libhpvpinn_testhooks.so $TOBJS & l2=$!
wait $l1; wait $l2
echo "built $(cd .. && pwd)/libhpvpinn.so (+ libhpvpinn_testhooks.so)"
