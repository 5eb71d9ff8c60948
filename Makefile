# `make` builds ./cudabrot (the drop-in CLI), cudabrot_amd/libcudabrot_amd.so (the C ABI) and the
# test oracle; `make ref` additionally builds the reference-derived checkers when /root/reference
# is present.
all:
	$(MAKE) -C cudabrot_amd/csrc all
	$(MAKE) -C oracle all

ref:
	$(MAKE) -C oracle ref

asm:
	$(MAKE) -C cudabrot_amd/csrc asm

clean:
	$(MAKE) -C cudabrot_amd/csrc clean
	$(MAKE) -C oracle clean

.PHONY: all ref asm clean
