set pagination off
set confirm off
info threads
thread apply all x/16i $pc-48
thread apply all info registers pc exec vcc s0 s1 s2 s3 s4 s5 s20 s21 s22 s23 s24 s25 s26 s27 s28 s29 s30 s31 s34 s35 s36 s37 s38 s44 s45 s76 s77 s88
thread apply all p/x $v0
thread apply all p/x $v1
thread apply all p/x $v2
thread apply all p/x $v3
thread apply all p/x $v4
thread apply all p/x $v5
thread apply all p/x $v6
thread apply all p/x $v7
thread apply all p/x $v46
thread apply all p/x $v47
thread apply all p/x $v48
thread apply all p/x $v49
quit
