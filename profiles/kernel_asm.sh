#!/bin/bash
# usage: profiles/kernel_asm.sh <file.s> <mangled-name-prefix>   -> the kernel's ISA on stdout (label .. .Lfunc_end)
awk -v pat="^$2" '$0 ~ pat && /:/ && !on {on=1} on {print} on && /^\.Lfunc_end/ {exit}' "$1"
