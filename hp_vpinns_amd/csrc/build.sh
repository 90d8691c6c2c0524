#!/bin/bash
# Build libhpvpinn.so for gfx950 in-tree (hipcc cross-compiles without a GPU), every stale object in parallel.
#   ../libhpvpinn.so            the product
#   ../libhpvpinn_testhooks.so  the same sources with -DHPV_TEST_HOOKS: the fault-injection knobs of the tests
#                               (HPV_DEBUG_SPLIT_SKIP: a partner workgroup stays away from an in-kernel exchange) exist ONLY
#                               there -- the product library neither reads that variable nor carries the branch in its kernels
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $HPV_EXTRA_FLAGS"   # e.g. HPV_EXTRA_FLAGS=-DHPV_FZ_TIMING
SRCS="kernels_generic kernels_mfma kernels_fused kernels_tall kernels_tile kernels_project hpv_api hpv_exchange hpv_bench"
ELEM_SHAPES="16,16,8,8 20,20,10,10 12,12,6,6"             # kernels_elem.hip: one object per element shape (= HPV_ELEM_SHAPES of hpv_mfma_dev.h)
WIDE_WIDTHS="24 32 40 48 64"                            # kernels_wide.hip: one object per hidden width (= HPV_WIDE_WIDTHS of hpv_mfma.h)
HOOKED="kernels_mfma kernels_fused kernels_tall hpv_api hpv_exchange"        # the sources that contain test hooks (built twice)
CHK="python3 ../../scripts/check_agpr.py"
# objects are cached by mtime; a change of flags must invalidate them (.flags remembers what the objects were built with)
if [ "$(cat .flags 2>/dev/null)" != "$FLAGS|$HPV_FUSED_EXTRA" ]; then rm -f *.o; echo "$FLAGS|$HPV_FUSED_EXTRA" > .flags; fi

stale() {   # stale <object> <source>: the object is missing or older than its source / any header
  [ ! -f "$1" ] && return 0
  for d in "$2" hpv_ctx.h hpv_internal.h hpv_mfma.h hpv_mfma_dev.h hpv_wide_dev.h hpv_project_wg.h hpv_math.h hpv_fused_dev.h ../../include/hpvpinn.h; do
    [ -f "$d" ] && [ "$d" -nt "$1" ] && return 0
  done
  return 1
}

# guard <asm> <hand-managed base> <mangled-name fragment>...: 0 clear, 1 tripped (some instantiation overlaps), 2 check impossible
guard() {
  local asm=$1 rc=0; shift
  while [ $# -ge 2 ]; do
    local r=0; $CHK $asm $1 $2 >&2 || r=$?
    [ $r -ge 2 ] && return 2
    [ $r -eq 1 ] && rc=1
    shift 2
  done
  return $rc
}

# Two sources park live values in hand-chosen AGPRs: they are compiled to assembly first and scripts/check_agpr.py verifies, per
# template instantiation, that the compiler's own registers stay clear of the hand-managed range.  If a compiler release ever
# needs more (exit 1), the library is STILL built -- without that kernel / instantiation (-DHPV_AGPR_GUARD_TRIPPED[_QT]: the
# launch functions decline, the callers fall back; hpv_build_info() reports it, bench.py prints it) -- and the build says so
# loudly.  If the check cannot run at all (exit 2: symbol not found after a rename, no assembly) the build FAILS.
compile_one() {   # compile_one <source stem> <object> <extra flags>
  local f=$1 obj=$2 extra=$3 XF="" asm=${2%.o}.s g=0
  if [ $f = kernels_fused ]; then
    XF="$HPV_FUSED_EXTRA"           # (A/B builds: flags for this file only, scripts/build_variant.sh --fused-only)
    $HIPCC $FLAGS $XF $extra -S --cuda-device-only $f.hip -o $asm 2>$asm.err || { cat $asm.err >&2; return 1; }
    # (instantiations: <L, SPLIT, QT, GS = false, element shape>: the GS = true ones hand-manage no registers; the hand-managed
    #  range starts at 256 - (tiles per wave - 2) x 10 L registers; the quarter-tile one sits closest to it and has its own fallback)
    local S=ELi20ELi20ELi10ELi10E
    g=0; guard $asm k_iter_fusedILi3ELb0ELb0ELb0${S} 106 k_iter_fusedILi3ELb1ELb0ELb0${S} 106 k_iter_fusedILi2ELb0ELb0ELb0${S} 156 k_iter_fusedILi2ELb1ELb0ELb0${S} 156 || g=$?
    [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check $f.hip" >&2; return 1; }
    if [ $g -eq 1 ]; then
      echo "build.sh: WARNING -- AGPR guard tripped in $f.hip: building without k_iter_fused (fallback = HPV_FUSE=b structure)" >&2
      XF="$XF -DHPV_AGPR_GUARD_TRIPPED"
    else
      g=0; guard $asm k_iter_fusedILi3ELb0ELb1ELb0${S} 106 k_iter_fusedILi2ELb0ELb1ELb0${S} 156 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the quarter-tile instantiation of $f.hip" >&2; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in the quarter-tile instantiation of k_iter_fused: building with 7 / 6 / 6 / 6 whole tiles per wave" >&2
        XF="$XF -DHPV_AGPR_GUARD_TRIPPED_QT"
      fi
      # the other element shapes (FZ_SHAPES of kernels_fused.hip): 16x16 / 8x8 (5 tiles per wave), 12x12 / 6x6 (3)
      local S16=ELi16ELi16ELi8ELi8E S12=ELi12ELi12ELi6ELi6E
      g=0; guard $asm k_iter_fusedILi3ELb0ELb0ELb0${S16} 166 k_iter_fusedILi3ELb1ELb0ELb0${S16} 166 k_iter_fusedILi2ELb0ELb0ELb0${S16} 196 k_iter_fusedILi2ELb1ELb0ELb0${S16} 196 \
                       k_iter_fusedILi3ELb0ELb1ELb0${S16} 166 k_iter_fusedILi2ELb0ELb1ELb0${S16} 196 \
                       k_iter_fusedILi3ELb0ELb0ELb0${S12} 226 k_iter_fusedILi3ELb1ELb0ELb0${S12} 226 k_iter_fusedILi3ELb0ELb1ELb0${S12} 226 \
                       k_iter_fusedILi2ELb0ELb0ELb0${S12} 236 k_iter_fusedILi2ELb1ELb0ELb0${S12} 236 k_iter_fusedILi2ELb0ELb1ELb0${S12} 236 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the extra element shapes of $f.hip" >&2; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in an extra element shape of k_iter_fused: those shapes run on the other structures" >&2
        XF="$XF -DHPV_FZ_NO_EXTRA_SHAPES"
      fi
    fi
  fi
  if [ $f = kernels_tall ]; then    # same hand-managed AGPR stash (4 tiles x L x 5 doubles at the top of the file)
    $HIPCC $FLAGS $extra -S --cuda-device-only $f.hip -o $asm 2>$asm.err || { cat $asm.err >&2; return 1; }
    # (template tail: <.., 80, 80, 5, 5, QT>; the QT instantiations keep one stash slot less: their range starts 30 registers higher)
    local T=ELi80ELi80ELi5ELi5
    g=0; guard $asm k_iter_tallILi2ELi1ELi3${T}ELb0 136 k_iter_tallILi2ELi0ELi3${T}ELb0 136 k_iter_tallILi2ELi1ELi2${T}ELb0 176 k_iter_tallILi2ELi0ELi2${T}ELb0 176 || g=$?
    [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check $f.hip" >&2; return 1; }
    if [ $g -eq 1 ]; then
      echo "build.sh: WARNING -- AGPR guard tripped in $f.hip: building without k_iter_tall (fallback = the separate launches)" >&2
      XF="$XF -DHPV_AGPR_GUARD_TRIPPED"
    else
      g=0; guard $asm k_iter_tallILi2ELi1ELi3${T}ELb1 166 k_iter_tallILi2ELi0ELi3${T}ELb1 166 k_iter_tallILi2ELi1ELi2${T}ELb1 196 k_iter_tallILi2ELi0ELi2${T}ELb1 196 || g=$?
      [ $g -eq 2 ] && { echo "build.sh: ERROR -- the AGPR guard could not check the quarter-tile instantiations of $f.hip" >&2; return 1; }
      if [ $g -eq 1 ]; then
        echo "build.sh: WARNING -- AGPR guard tripped in the quarter-tile instantiations of k_iter_tall: building with whole tiles only" >&2
        XF="$XF -DHPV_AGPR_GUARD_TRIPPED_QT"
      fi
    fi
  fi
  $HIPCC $FLAGS $XF $extra -c $f.hip -o $obj
}

pids=(); names=()
for f in $SRCS; do
  [ -f $f.hip ] || continue
  if stale $f.o $f.hip; then compile_one $f $f.o "" & pids+=($!); names+=($f.o); fi
done
for f in $HOOKED; do
  if stale $f.th.o $f.hip; then compile_one $f $f.th.o "-DHPV_TEST_HOOKS" & pids+=($!); names+=($f.th.o); fi
done
for w in $WIDE_WIDTHS; do
  if stale kernels_wide_$w.o kernels_wide.hip; then
    $HIPCC $FLAGS -DHPV_WIDE_H=$w -c kernels_wide.hip -o kernels_wide_$w.o & pids+=($!); names+=(kernels_wide_$w.o)
  fi
done
for sh in $ELEM_SHAPES; do
  IFS=, read qx qy ntx nty <<< "$sh"
  o=kernels_elem_${qx}_${qy}_${ntx}_${nty}.o
  if stale $o kernels_elem.hip; then
    $HIPCC $FLAGS -DHPV_ELEM_QX=$qx -DHPV_ELEM_QY=$qy -DHPV_ELEM_NTX=$ntx -DHPV_ELEM_NTY=$nty -c kernels_elem.hip -o $o & pids+=($!); names+=($o)
  fi
done
fail=0
for i in "${!pids[@]}"; do
  if ! wait ${pids[$i]}; then echo "build.sh: ERROR -- ${names[$i]} failed" >&2; rm -f ${names[$i]}; fail=1; fi
done
[ $fail -eq 0 ] || exit 1

OBJS=""; TOBJS=""
for f in $SRCS; do
  [ -f $f.hip ] || continue
  OBJS="$OBJS $f.o"
  case " $HOOKED " in *" $f "*) TOBJS="$TOBJS $f.th.o";; *) TOBJS="$TOBJS $f.o";; esac
done
for w in $WIDE_WIDTHS; do OBJS="$OBJS kernels_wide_$w.o"; TOBJS="$TOBJS kernels_wide_$w.o"; done
for sh in $ELEM_SHAPES; do o=kernels_elem_${sh//,/_}.o; OBJS="$OBJS $o"; TOBJS="$TOBJS $o"; done
# -Bsymbolic: the two libraries may live in one process (the tests load both); each must bind its internal calls to itself
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../libhpvpinn.so $OBJS
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../libhpvpinn_testhooks.so $TOBJS
echo "built $(cd .. && pwd)/libhpvpinn.so (+ libhpvpinn_testhooks.so)"
