# Sourced by the experiment scripts that rebuild the engine IN PLACE on the GPU box (round-2 advice: a failed build must not be benched as
# if it were the variant, and an interrupted script must not leave an experimental build installed).
#   pt_make <make arguments...>   runs make quietly; on failure prints the log and exits
#   the first call installs a trap that restores the product build (plain `make all`) when the script ends, however it ends
PT_BUILD_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
pt_restore_build() { make -j8 -C "$PT_BUILD_ROOT/rust-pathtracer_amd/csrc" clean > /dev/null 2>&1; make -j8 -C "$PT_BUILD_ROOT/rust-pathtracer_amd/csrc" all > /dev/null 2>&1 || echo "WARNING: the product build could not be restored"; }
pt_make() {
  if [ -z "${PT_BUILD_TRAP:-}" ]; then PT_BUILD_TRAP=1; trap pt_restore_build EXIT; fi
  local log; log=$(mktemp)
  if ! make "$@" > "$log" 2>&1; then echo "BUILD FAILED: make $*"; tail -30 "$log"; rm -f "$log"; exit 1; fi
  rm -f "$log"
}
pt_check_build() { if [ "$1" -ne 0 ]; then echo "BUILD FAILED: $2"; exit 1; fi; }
