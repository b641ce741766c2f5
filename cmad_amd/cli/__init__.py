"""`cmad`-compatible command line over the device evaluator (material-point decks)."""
