"""Stand-in for the third-party ``katsdpsigproc`` package (absent here).

Only what the reference's *host* classes touch at import/run time.  Device
classes are empty placeholders: the reference GPU path is never executed.
"""
