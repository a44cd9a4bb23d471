# Convenience targets; the Python entry points (`__graft_entry__.build()`, `python -m spectrograms_amd.build`) run the same
# commands (per-file objects under build/obj, linked into the in-tree library).  The library is plain hipcc output: a Rust
# `build.rs` of the reference crate's `hip` feature would do the same.
.PHONY: all lib oracle test-cpu clean
all: lib oracle

lib:
	python -m spectrograms_amd.build

oracle:
	$(MAKE) -C oracle

test-cpu: all
	python -m pytest tests -q -m "not gpu"

clean:
	rm -rf build spectrograms_amd/libspectro_hip.so
	$(MAKE) -C oracle clean || true
