# convenience targets; the authoritative build is __graft_entry__.build()
all:
	$(MAKE) -C multigrid_petsc_amd/csrc all
	$(MAKE) -C oracle all
test-cpu: all
	python -m pytest tests -x -q -m "not gpu"
test-gpu: all
	python -m pytest tests -x -q -m gpu
clean:
	$(MAKE) -C multigrid_petsc_amd/csrc clean
	$(MAKE) -C oracle clean
.PHONY: all test-cpu test-gpu clean
