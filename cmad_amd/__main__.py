"""`python -m cmad_amd <subcommand> deck.yaml` == the reference's `cmad <subcommand> deck.yaml`."""
import sys

from .cli.main import main

sys.exit(main())
