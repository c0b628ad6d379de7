"""`python -m src.run_modegpt --model ... --compression_ratio 0.3 --order mlp,qk,vo ...` (README.md:31-33 of the
reference) -> modegpt_amd.run_modegpt.main."""
from modegpt_amd.run_modegpt import main  # noqa: F401

if __name__ == "__main__":
    main()
