// maps.S -- the interior map (cells of the c-plane whose samples provably never escape: tools/interior_map.c), embedded
// in the library and the binary (capi.hip, device_map): no file to find, lose or swap at run time.  `make` unpacks the
// kept ../interior_map.bin.gz into build/ and checks its sha256 against the digest kept in the tree
// (map_digests.sha256) before this file is assembled.
        .section .rodata
        .balign 64
        .globl cb_embedded_interior_map
        .globl cb_embedded_interior_map_end
cb_embedded_interior_map:
        .incbin "build/interior_map.bin"
cb_embedded_interior_map_end:
        .section .note.GNU-stack,"",@progbits
