#!/bin/bash
# waitmap.sh <asm> <mangled-substring>: loads / stores / scratch / barriers / vmcnt(0) waits of one kernel in program order
awk -v k="$2" '$0 ~ "^_Z" && index($0,k) && /:/ {f=1} f{print} f&&/\.Lfunc_end/{exit}' "$1" > /tmp/k.s
grep -n "s_barrier\|s_waitcnt vmcnt\|global_store\|global_load\|scratch_\|Loop Header" /tmp/k.s | awk -F: '{print $1": "$2}' | sed 's/\s\+/ /g' | awk '{ if ($0 ~ /global_load/) l++; else if ($0 ~ /global_store/) s++; else if ($0 ~ /scratch_load/) sl++; else if ($0 ~ /scratch_store/) ss++; else { if (l||s||sl||ss) print "   ... loads", l+0, "stores", s+0, "scratch ld/st", sl+0, ss+0; l=0; s=0; sl=0; ss=0; print } } END{print "   ... loads", l+0, "stores", s+0, "scratch ld/st", sl+0, ss+0}'
