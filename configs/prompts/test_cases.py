"""`from configs.prompts.test_cases import TestCasesDict` (scripts/pose2vid.py:19) -- the module is
missing from the reference tree; an empty mapping keeps the script importable."""
TestCasesDict = {}
