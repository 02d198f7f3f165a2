#!/usr/bin/env bash
# The cross-check against Kmers.jar (needs a JVM) uses the CPU restatements as file readers, so it lives with the
# other checkers: tests/jar_crosscheck.sh.  Same arguments.
exec "$(cd "$(dirname "$0")/.." && pwd)/tests/jar_crosscheck.sh" "$@"
