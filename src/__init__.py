"""Import-path alias: ``src.*`` of the reference (ltdung/WHVI) resolves to this repo's ``whvi_amd`` package, so
code and tests written against the reference (``from src.layers import WHVILinear``) run on the MI355X
implementation without edits.  One re-export per reference module; no logic lives here."""
